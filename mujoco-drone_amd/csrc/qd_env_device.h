// qd_env_device.h -- device-side building blocks shared by the translation units of libqd.so: the arena layout, the
// kernel argument block, one env in registers and its plane loads / stores, the reset pool, observation-row staging,
// and the LDS hand-over of the cooperative (multi-wave) step.  Moved out of qd_kernels.hip unchanged so that the persistent
// fragment kernel (qd_rollout_coop.hip) compiles on its own.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../../include/qd.h"
#include "qd_dynamics.h"
#include "qd_math.h"
#include "qd_model.h"
#include "qd_obsrew.h"
#include "qd_pid.h"
#include "qd_rng.h"

namespace qd {

enum Group {
  G_POS = 0, G_QUAT, G_VEL, G_ANG, G_ACT, G_AUX, G_ACC, G_M0, G_M1, G_M2, G_M3, G_M4, G_M5, G_M6, G_P0, G_P1, G_REF,
  G_NX0, G_NX1, G_NX2, G_NX3, G_NX4,  // reset pool, slot 0: pre-sampled initial state of the env's episodes with an EVEN counter
  G_NY0, G_NY1, G_NY2, G_NY3, G_NY4,  //             slot 1: ... with an ODD counter (filled by sampler waves, see "reset pool")
  G_NXA0, G_NXA1, G_NXA2, G_NXA3,     // slot 0: the new episode's first accelerometer reading as c0 + sum a_i col_i (15 floats), for
  G_NYA0, G_NYA1, G_NYA2, G_NYA3,     // slot 1: configurations whose observation row carries the sensor (obs_needs_acc)
  G_C0, G_C1, G_C2, G_C3,             // memory of the analytic PID cascade (qd_pid.h); touched only by the qd_pid_* entry points
  NUM_GROUPS
};
// float64 planes behind the float4 groups: the six raw parameters (the 5-digit XML rounding must see float64 values), then what the
// floor-contact step needs of a parameter set and used to re-derive in every substep: the nine %.5g-rounded geom sizes /
// placements, the reach below the origin, the translational body_invweight0 of the three bodies (k_floor_consts, qd_step_floor.hip)
constexpr int RAW_PARAMS = 6;
constexpr int RAW_PLANES = RAW_PARAMS + FLOOR_CONSTS;   // FLOOR_CONSTS and its indices: qd_model.h
constexpr int POOL_STAT_WORDS = 64;   // behind the refill counters: [0] in-kernel resets served by the pool, [1] sampled inline (qd_pool_counters);
                                      // [8 + k]: events that must not happen (qd_health_counters): k = 0 a bounded in-kernel poll ran out
constexpr int PAD = 256;

struct KArgs {
  float4* g;
  double* raw;
  uint32_t* need;     // [npad / 64] reset-pool refill requests, one counter per 64 consecutive envs
  int npad, n;
  float ref[4];
  int per_env_ref;
  float h;
  int frame_skip, ctrl_map, obs_kind, reward_kind, term_kind;
  float max_distance;
  int max_steps, auto_reset, D;
  int obs_needs_acc;  // the observation variant reads the accelerometer entries of the state vector
  int use_pool;       // auto_reset with random starts: sampler workgroups keep the reset pool filled
  int main_blocks;    // workgroups [0, main_blocks) step envs; [main_blocks, 2*main_blocks) are samplers
  int ref_mode;       // QD_REF_CIRCLE: the reference is a function of (env, episode step), see moving_reference()
  float ref_radius, ref_omega_dt, ref_phase_step;  // radius, 2 pi f dt, 2 pi / N
  int ref_k0, ref_kmax;                            // QD_REF_STEP / RAMP: first sample with t_k >= t0; last sample index
  float ref_dt, ref_t0, ref_inv_span, ref_end[4];  // ramp: (k dt - t0) / (duration - t0)
  unsigned long long seed;
  SampleCfg sc;
};

struct EnvRegs {  // everything one lane keeps in registers for one env
  State<float> s;
  Model<float> M;
  V3<float> acc;
  float par[6];
  float ref[4];
  int num_steps;
  uint32_t episode;
  uint32_t flags;  // FLAG_ACC_STALE: the stored accelerometer value predates an in-kernel reset
};
constexpr uint32_t FLAG_ACC_STALE = 1u;

// one lane asks for a refill of its env's reset-pool entries (see "reset pool"; the counter covers the lane's 64-env group)
// (the counters are addressed as GLOBAL memory explicitly: a KArgs rebuilt from dwords -- step_kargs -- has lost the address space
// of its pointers, and a FLAT atomic also counts on the LDS counter, so the next workgroup barrier would wait for it)
typedef __attribute__((address_space(1))) uint32_t* need_ptr;
__device__ __forceinline__ void pool_request(const KArgs& a, int i) {
  __hip_atomic_fetch_add((need_ptr)a.need + (i >> 6), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// bookkeeping of how in-kernel resets got their state (a handful of atomics per launch: only truncating lanes come here)
__device__ __forceinline__ void pool_count(const KArgs& a, bool taken) {
  __hip_atomic_fetch_add((need_ptr)a.need + (a.npad >> 6) + (taken ? 0 : 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// an event that must not happen, counted where the host can see it (qd_health_counters)
__device__ __forceinline__ void health_count(const KArgs& a, int which) {
  __hip_atomic_fetch_add((need_ptr)a.need + (a.npad >> 6) + 8 + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Compile-time specialisations of the fused step for the configurations the reference trains with; every other
// configuration runs the generic instantiation, which dispatches on the KArgs fields at run time (wave-uniform
// branches).  Specialising removes the dispatch chains and lets the compiler drop state-vector entries the
// selected observation / reward never reads.
enum Spec {
  SPEC_GENERIC = 0,
  SPEC_RMA = 1,     // train_PPO.py / train_RMA.py: LocalFrameRPYParamsEnv + distance_energy_reward      (BASELINE cfg 3, 4)
  SPEC_LSTM = 2,    // train_LSTM.py: LocalFrameFullStateEnv + distance_energy_reward_pendulum_en4          (BASELINE cfg 5)
  SPEC_SIMPLE = 3,  // SimpleDrone.py: 6-value observation, drone-0 style reward / termination, direct ctrl (BASELINE cfg 1, 2)
  SPEC_GENERIC_FS1 = 4,  // any observation / reward (run-time dispatch) with the usual skip_steps = 1 fixed at compile time
  SPEC_FLOOR = 5         // SPEC_GENERIC of the single-body model with the floor contact (qd_contact.h), qd_config.floor_contact
};
template <int SPEC> constexpr bool spec_runtime() { return SPEC == SPEC_GENERIC || SPEC == SPEC_GENERIC_FS1 || SPEC == SPEC_FLOOR; }
template <int SPEC> __device__ __forceinline__ int spec_obs(const KArgs& a) {
  return SPEC == SPEC_RMA ? (int)OBS_RPY_PARAMS : SPEC == SPEC_LSTM ? (int)OBS_FULLSTATE : SPEC == SPEC_SIMPLE ? (int)OBS_SIMPLE : a.obs_kind;
}
template <int SPEC> __device__ __forceinline__ int spec_reward(const KArgs& a) {
  return SPEC == SPEC_RMA ? (int)REW_DISTANCE_ENERGY : SPEC == SPEC_LSTM ? (int)REW_PEND_EN4 : SPEC == SPEC_SIMPLE ? (int)REW_SIMPLE : a.reward_kind;
}
template <int SPEC> __device__ __forceinline__ int spec_term(const KArgs& a) {
  return SPEC == SPEC_SIMPLE ? (int)QD_TERM_SIMPLE : spec_runtime<SPEC>() ? a.term_kind : (int)QD_TERM_DEFAULT;
}
// physics substeps per env step: compile-time in the specialisations (a run-time loop keeps the whole model and the
// controls alive across the float64 core: +46 registers, every one of them an AGPR copy per use)
template <int SPEC> constexpr int spec_frame_skip() { return SPEC == SPEC_SIMPLE ? 2 : (SPEC == SPEC_GENERIC || SPEC == SPEC_FLOOR) ? 0 : 1; }
template <int SPEC> __device__ __forceinline__ int spec_ctrl(const KArgs& a) {
  return SPEC == SPEC_SIMPLE ? (int)QD_CTRL_DIRECT : spec_runtime<SPEC>() ? a.ctrl_map : (int)QD_CTRL_AFFINE;
}

// the waypoint generators of evaluation.py:135-152 evaluated in the kernel, k = the env's episode step:
// circle around the configured reference (one phase per env), step and ramp from the reference to ref_end
__device__ __forceinline__ void moving_reference(const KArgs& a, int i, int k, float ref[4]) {
  if (a.ref_mode == QD_REF_CIRCLE) {
    float sn, cs;
    qsincos(a.ref_omega_dt * (float)k + a.ref_phase_step * (float)i, &sn, &cs);
    ref[0] = a.ref[0] + a.ref_radius * cs;
    ref[1] = a.ref[1] + a.ref_radius * sn;
    ref[2] = a.ref[2];
    ref[3] = a.ref[3];
  } else {
    const int kk = min(k, a.ref_kmax);
    float w = 0.f;
    if (kk >= a.ref_k0) w = a.ref_mode == QD_REF_STEP ? 1.f : ((float)kk * a.ref_dt - a.ref_t0) * a.ref_inv_span;
#pragma unroll
    for (int c = 0; c < 4; c++) ref[c] = a.ref_mode == QD_REF_STEP && w == 1.f ? a.ref_end[c] : a.ref[c] + w * (a.ref_end[c] - a.ref[c]);
  }
}

// WITH_ACC = false: the stored accelerometer reading is not fetched (the step kernels overwrite it).
// FOLD = true: planes M3..M6 (fluid coefficients) are not fetched but re-folded from M0..M2 in float32; used by
// the 256-thread (HBM-bound) step kernels, where 64 bytes less per env-step matter more than ~90 instructions.
template <bool LOAD, bool WITH_ACC = true, bool FOLD = false>
__device__ __forceinline__ void load_env_planes(const float4* g, int np, int i, EnvRegs& e) {
  const float4 pos = g[G_POS * np + i], qt = g[G_QUAT * np + i], vel = g[G_VEL * np + i], ang = g[G_ANG * np + i];
  const float4 act = g[G_ACT * np + i], aux = g[G_AUX * np + i];
  const float4 acc = WITH_ACC ? g[G_ACC * np + i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 m0 = g[G_M0 * np + i], m1 = g[G_M1 * np + i], m2 = g[G_M2 * np + i];
  const float4 p0 = g[G_P0 * np + i], p1 = g[G_P1 * np + i];
  e.s.px = pos.x; e.s.py = pos.y; e.s.pz = pos.z; e.s.th1 = pos.w;
  e.s.qw = qt.x; e.s.qx = qt.y; e.s.qy = qt.z; e.s.qz = qt.w;
  e.s.vx = vel.x; e.s.vy = vel.y; e.s.vz = vel.z; e.s.th2 = vel.w;
  e.s.wx = ang.x; e.s.wy = ang.y; e.s.wz = ang.z; e.s.thd1 = ang.w;
  e.s.a0 = act.x; e.s.a1 = act.y; e.s.a2 = act.z; e.s.a3 = act.w;
  e.s.thd2 = aux.x; e.num_steps = __float_as_int(aux.y); e.episode = __float_as_uint(aux.z); e.flags = __float_as_uint(aux.w);
  e.acc = mk<float>(acc.x, acc.y, acc.z);
  e.M.m0 = m0.x; e.M.c0z = m0.y; e.M.I0x = m0.z; e.M.I0y = m0.w;
  e.M.I0z = m1.x; e.M.rot = m1.y; e.M.gearF = m1.z; e.M.gearT = m1.w;
  e.M.inv_tau = m2.x; e.M.m2 = m2.y; e.M.lc = m2.z; e.M.I2t = m2.w;
  e.M.I2a = p1.z;  // the raw-parameter plane P1 carries a copy of I2a in its spare slot
  e.M.klin2 = e.M.kang2 = e.M.qlt2 = e.M.qla2 = e.M.qat2 = e.M.qaa2 = e.M.pad = 0.f;
  if (FOLD) {
    fluid_coeffs_inline(e.M.I0x, e.M.I0y, e.M.I0z, e.M.m0, &e.M.klin0, &e.M.kang0, &e.M.qlx0, &e.M.qly0, &e.M.qlz0, &e.M.qax0,
                        &e.M.qay0, &e.M.qaz0);
    if (LOAD) {
      float qly, qay;
      fluid_coeffs_inline(e.M.I2t, e.M.I2t, e.M.I2a, e.M.m2, &e.M.klin2, &e.M.kang2, &e.M.qlt2, &qly, &e.M.qla2, &e.M.qat2, &qay,
                          &e.M.qaa2);
    }
  } else {
    const float4 m3 = g[G_M3 * np + i], m4 = g[G_M4 * np + i];
    e.M.klin0 = m3.y; e.M.kang0 = m3.z; e.M.qlx0 = m3.w;
    e.M.qly0 = m4.x; e.M.qlz0 = m4.y; e.M.qax0 = m4.z; e.M.qay0 = m4.w;
    const float4 m5 = g[G_M5 * np + i];
    e.M.qaz0 = m5.x;
    if (LOAD) {
      const float4 m6 = g[G_M6 * np + i];
      e.M.klin2 = m5.y; e.M.kang2 = m5.z; e.M.qlt2 = m5.w;
      e.M.qla2 = m6.x; e.M.qat2 = m6.y; e.M.qaa2 = m6.z;
    }
  }
  e.par[0] = p0.x; e.par[1] = p0.y; e.par[2] = p0.z; e.par[3] = p0.w; e.par[4] = p1.x; e.par[5] = p1.y;
}
// the env's reference point: the one thing of its registers that depends on launch arguments other than the arena
__device__ __forceinline__ void load_env_ref(const KArgs& a, int i, EnvRegs& e) {
  if (a.ref_mode != QD_REF_STATIC) {
    moving_reference(a, i, e.num_steps, e.ref);
  } else if (a.per_env_ref) {
    const float4 r = a.g[G_REF * a.npad + i];
    e.ref[0] = r.x; e.ref[1] = r.y; e.ref[2] = r.z; e.ref[3] = r.w;
  } else {
    e.ref[0] = a.ref[0]; e.ref[1] = a.ref[1]; e.ref[2] = a.ref[2]; e.ref[3] = a.ref[3];
  }
}

template <bool LOAD, bool WITH_ACC = true, bool FOLD = false>
__device__ __forceinline__ void load_env(const KArgs& a, int i, EnvRegs& e) {
  load_env_planes<LOAD, WITH_ACC, FOLD>(a.g, a.npad, i, e);
  load_env_ref(a, i, e);
}

// everything of store_env but the accelerometer plane
__device__ __forceinline__ void store_env_state(const KArgs& a, int i, const EnvRegs& e) {
  float4* g = a.g;
  const int np = a.npad;
  g[G_POS * np + i] = make_float4(e.s.px, e.s.py, e.s.pz, e.s.th1);
  g[G_QUAT * np + i] = make_float4(e.s.qw, e.s.qx, e.s.qy, e.s.qz);
  g[G_VEL * np + i] = make_float4(e.s.vx, e.s.vy, e.s.vz, e.s.th2);
  g[G_ANG * np + i] = make_float4(e.s.wx, e.s.wy, e.s.wz, e.s.thd1);
  g[G_ACT * np + i] = make_float4(e.s.a0, e.s.a1, e.s.a2, e.s.a3);
  g[G_AUX * np + i] = make_float4(e.s.thd2, __int_as_float(e.num_steps), __uint_as_float(e.episode), __uint_as_float(e.flags));
}
__device__ __forceinline__ void store_env(const KArgs& a, int i, const EnvRegs& e) {
  float4* g = a.g;
  const int np = a.npad;
  g[G_POS * np + i] = make_float4(e.s.px, e.s.py, e.s.pz, e.s.th1);
  g[G_QUAT * np + i] = make_float4(e.s.qw, e.s.qx, e.s.qy, e.s.qz);
  g[G_VEL * np + i] = make_float4(e.s.vx, e.s.vy, e.s.vz, e.s.th2);
  g[G_ANG * np + i] = make_float4(e.s.wx, e.s.wy, e.s.wz, e.s.thd1);
  g[G_ACT * np + i] = make_float4(e.s.a0, e.s.a1, e.s.a2, e.s.a3);
  g[G_AUX * np + i] = make_float4(e.s.thd2, __int_as_float(e.num_steps), __uint_as_float(e.episode), __uint_as_float(e.flags));
  g[G_ACC * np + i] = make_float4(e.acc.x, e.acc.y, e.acc.z, 0.f);
}

// accelerometer refresh = the part of mj_forward the reference observes after set_state
template <bool LOAD>
__device__ __forceinline__ void refresh_sensor(const KArgs& a, EnvRegs& e) {
  Accel<float> ex, im;
  forward<float, LOAD>(e.M, e.s, a.h, &ex, &im, &e.acc);
  e.flags &= ~FLAG_ACC_STALE;
}

// the initial state of episode `episode` of env i (positions, attitude, velocities, hinges; NOT the activations)
template <bool LOAD>
__device__ __forceinline__ void sample_episode(const KArgs& a, int i, uint32_t episode, State<float>& s) {
  if (!LOAD && a.sc.random_start == QD_START_SIMPLE) {   // SimpleDrone.reset_model: the no-load model only (qd_create enforces it)
    sample_simple(a.sc, a.seed, (uint32_t)i, (uint32_t)a.n, episode, s);
  } else {
    float z[16], u[2];
    sample_draws(a.seed, (uint32_t)i, episode, z, u);
    sample_state<LOAD>(a.sc, z, u, s);
  }
}

// ---- reset pool ------------------------------------------------------------------------------
// Drawing a new initial state costs ~1000 instructions (5 Philox blocks, 8 Box-Muller pairs, the
// transforms).  Done inline by the lanes that truncate, it sits on the critical path of their
// wavefront -- and with 4096 envs and ~150-step episodes a third of the 64 wavefronts contain such a
// lane every step, so the whole launch waits for it.  Instead the step kernels are launched with extra
// "sampler" workgroups (on CUs the physics waves leave idle) that keep the initial state of each env's
// NEXT episode ready in the arena: episode e lives in slot e & 1 (planes NX* / NY*); a truncating lane only
// loads its entry.  The sample for (env, episode) is a pure function of the Philox counter, so results do
// not depend on who computes it or when.  Protocol (no intra-launch synchronisation):
//   explicit reset (k_reset)  : draws episode e inline, leaves sample(e+1) in its slot for the new counter e+1, adds 1 to need[w]
//   sampler, any step launch  : if need[w] == 0 for its 64 envs -> exit.  Otherwise, with E = AUX.episode as it reads it,
//                               makes slot (E+1) & 1 hold sample(E+1) and subtracts the request count it had read
//   physics, on truncation    : in episode e, if slot e & 1 is valid with tag e -> take it, mark it consumed and add 1
//                               to need[w]; otherwise (never filled, pool off) sample inline.
// A sampler only ever writes the slot of the episode AFTER the counter it reads, a lane only reads the slot of its
// counter at launch start, and it stores a new counter after it is done reading: the slot a lane may read is never the
// one a sampler of the same launch writes.  An entry is therefore complete one kernel boundary before it can be read --
// no fence, and its planes may be fetched in any order and ahead of time (k_step_coop).  A lost race costs speed only:
// a request counted twice re-checks 64 tags, an entry that is not there is sampled inline.
__device__ __forceinline__ int pool_slot(uint32_t episode) { return (episode & 1u) ? (int)G_NY0 : (int)G_NX0; }
__device__ __forceinline__ int pool_acc_slot(uint32_t episode) { return (episode & 1u) ? (int)G_NYA0 : (int)G_NXA0; }
// entry stages (word z of the tag plane): 0 empty / consumed, 1 state sampled, 2 state + the accelerometer's affine form
constexpr uint32_t POOL_STATE = 1u, POOL_FULL = 2u;

// the 15 floats of (c0, col[0..3]) in four planes, and the reading they give for activations a
struct SensorAffine { float4 p[4]; };
__device__ __forceinline__ V3<float> sensor_from_affine(const SensorAffine& sa, float a0, float a1, float a2, float a3) {
  const float4 p0 = sa.p[0], p1 = sa.p[1], p2 = sa.p[2], p3 = sa.p[3];
  // planes: (c0.xyz, col0.x) (col0.yz, col1.xy) (col1.z, col2.xyz) (col3.xyz, -)
  return mk<float>(p0.x + a0 * p0.w + a1 * p1.z + a2 * p2.y + a3 * p3.x,
                   p0.y + a0 * p1.x + a1 * p1.w + a2 * p2.z + a3 * p3.y,
                   p0.z + a0 * p1.y + a1 * p2.x + a2 * p2.w + a3 * p3.z);
}

__device__ __forceinline__ void pool_store(const KArgs& a, int i, uint32_t episode, const State<float>& s) {
  float4* g = a.g;
  const int np = a.npad, base = pool_slot(episode);
  g[(base + 0) * np + i] = make_float4(s.px, s.py, s.pz, s.th1);
  g[(base + 1) * np + i] = make_float4(s.qw, s.qx, s.qy, s.qz);
  g[(base + 2) * np + i] = make_float4(s.vx, s.vy, s.vz, s.th2);
  g[(base + 3) * np + i] = make_float4(s.wx, s.wy, s.wz, s.thd1);
  g[(base + 4) * np + i] = make_float4(s.thd2, __uint_as_float(episode), __uint_as_float(POOL_STATE), 0.f);
}

// Bring the slot of episode `next` one stage closer to complete; returns true once it is.  Stage 1 draws the state (~1000
// instructions); stage 2 -- only for configurations whose observation row carries the accelerometer -- evaluates the sensor's
// affine form at that state (~1300).  One stage per call: an entry is prepared an episode ahead, so there is no hurry, and a
// sampler wave that did both in one launch would outlast the physics waves it is meant to hide behind.
template <bool LOAD>
__device__ __forceinline__ bool pool_fill(const KArgs& a, int i, uint32_t next, float4 nx4 /* the slot's tag plane */) {
  float4* g = a.g;
  const int np = a.npad, base = pool_slot(next);
  const uint32_t stage = __float_as_uint(nx4.y) == next ? __float_as_uint(nx4.z) : 0u;
  const uint32_t want = (LOAD && a.obs_needs_acc) ? POOL_FULL : POOL_STATE;
  if (stage >= want) return true;
  if (stage == 0u) {
    State<float> s;
    sample_episode<LOAD>(a, i, next, s);
    pool_store(a, i, next, s);
    return want == POOL_STATE;
  }
  if constexpr (LOAD) {
    EnvRegs e;
    load_env<true, false, false>(a, i, e);   // for the model; the state is the pre-sampled one
    const float4 p = g[(base + 0) * np + i], q = g[(base + 1) * np + i], v = g[(base + 2) * np + i], w = g[(base + 3) * np + i];
    e.s.px = p.x; e.s.py = p.y; e.s.pz = p.z; e.s.th1 = p.w;
    e.s.qw = q.x; e.s.qx = q.y; e.s.qy = q.z; e.s.qz = q.w;
    e.s.vx = v.x; e.s.vy = v.y; e.s.vz = v.z; e.s.th2 = v.w;
    e.s.wx = w.x; e.s.wy = w.y; e.s.wz = w.z; e.s.thd1 = w.w;
    e.s.thd2 = nx4.x;
    V3<float> c0, col[4];
    sensor_affine<float>(e.M, e.s, a.h, &c0, col);
    const int ab = pool_acc_slot(next);
    g[(ab + 0) * np + i] = make_float4(c0.x, c0.y, c0.z, col[0].x);
    g[(ab + 1) * np + i] = make_float4(col[0].y, col[0].z, col[1].x, col[1].y);
    g[(ab + 2) * np + i] = make_float4(col[1].z, col[2].x, col[2].y, col[2].z);
    g[(ab + 3) * np + i] = make_float4(col[3].x, col[3].y, col[3].z, 0.f);
    g[(base + 4) * np + i] = make_float4(nx4.x, nx4.y, __uint_as_float(POOL_FULL), 0.f);
  }
  return true;
}
template <bool LOAD>
__device__ __forceinline__ bool pool_fill(const KArgs& a, int i, uint32_t next) {
  return pool_fill<LOAD>(a, i, next, a.g[(pool_slot(next) + 4) * a.npad + i]);
}

// sample_state for this lane's env, episode counter advanced.  eager_sensor: run mj_forward's sensor
// part now (reset kernels); otherwise only mark the stored reading stale -- it is recomputed by the
// next physics step anyway, and by the state/observation getters if they run before that step.
template <bool LOAD>
__device__ __forceinline__ void resample(const KArgs& a, int i, EnvRegs& e, bool eager_sensor) {
  sample_episode<LOAD>(a, i, e.episode, e.s);
  e.episode += 1u;
  e.num_steps = 0;
  // the new counter's own entry (samplers only ever fill the one after it); both stages: a reset kernel is not in a hurry.
  // And a request, so that the next step launch's samplers prepare the episode after this one: without it the env's second
  // truncation after every explicit reset found no entry and sampled inline (one in seven resets of the regen-every-1024 bench).
  if (a.use_pool) {
    if (!pool_fill<LOAD>(a, i, e.episode)) pool_fill<LOAD>(a, i, e.episode);
    pool_request(a, i);
  }
  if (eager_sensor) refresh_sensor<LOAD>(a, e);
  else e.flags |= FLAG_ACC_STALE;
}

// body of a sampler wavefront: its 64 lanes serve the envs [j0, j0 + 64) of one refill counter
template <bool LOAD>
__device__ __forceinline__ void sampler_wave(const KArgs& a, int j) {
  const int w = j >> 6;
  if ((w << 6) >= a.n) return;
  const uint32_t pending = __hip_atomic_load((need_ptr)a.need + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (pending == 0u) return;
  bool done = true;
  if (j < a.n) {
    // the counter and both slots' tag planes in ONE round trip (which slot is the next episode's depends on the counter): a
    // refilling sampler wave is about as long as a physics wave, and a dependent load here is ~1000 cycles of it
    const float4 aux = a.g[G_AUX * a.npad + j], tx = a.g[G_NX4 * a.npad + j], ty = a.g[G_NY4 * a.npad + j];
    const uint32_t next = __float_as_uint(aux.z) + 1u;
    done = pool_fill<LOAD>(a, j, next, (next & 1u) ? ty : tx);
  }
  // entries that still lack a stage keep the request alive: the next launch's sampler comes back for them
  if (__all(done ? 1 : 0) && (threadIdx.x & 63) == 0) __hip_atomic_fetch_sub((need_ptr)a.need + w, pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what a reset does to everything but the sampled state (shared by the single-wave and the cooperative step)
__device__ __forceinline__ void reset_bookkeeping(State<float>& s, uint32_t& episode, int& num_steps) {
  // activations survive a reset (reference quirk C-2) -- unless they diverged: MuJoCo's bad-state check would have
  // called mj_resetData, which zeroes them
  if (!(fabsf(s.a0) + fabsf(s.a1) + fabsf(s.a2) + fabsf(s.a3) < 1e10f)) s.a0 = s.a1 = s.a2 = s.a3 = 0.f;
  episode += 1u;
  num_steps = 0;
}

// reset of a truncated lane inside the step kernel: pool entry if it is there, inline sampling otherwise
// POOL = false: instantiations that are never launched with sampler workgroups (the 256-thread step kernels of large
// batches) carry no pool code
template <bool LOAD, bool POOL = true>
__device__ __forceinline__ void reset_in_step(const KArgs& a, int i, EnvRegs& e) {
  bool taken = false, have_sa = false;
  SensorAffine sa;
  if (POOL && a.use_pool) {
    float4* g = a.g;
    const int np = a.npad, base = pool_slot(e.episode);
    const float4 nx4 = g[(base + 4) * np + i];
    if (__float_as_uint(nx4.z) != 0u && __float_as_uint(nx4.y) == e.episode) {
      const float4 p = g[(base + 0) * np + i], q = g[(base + 1) * np + i], v = g[(base + 2) * np + i], w = g[(base + 3) * np + i];
      e.s.px = p.x; e.s.py = p.y; e.s.pz = p.z; e.s.th1 = p.w;
      e.s.qw = q.x; e.s.qy = q.z; e.s.qx = q.y; e.s.qz = q.w;
      e.s.vx = v.x; e.s.vy = v.y; e.s.vz = v.z; e.s.th2 = v.w;
      e.s.wx = w.x; e.s.wy = w.y; e.s.wz = w.z; e.s.thd1 = w.w;
      e.s.thd2 = nx4.x;
      if (LOAD && a.obs_needs_acc && __float_as_uint(nx4.z) == POOL_FULL) {
        const int ab = pool_acc_slot(e.episode);
#pragma unroll
        for (int k = 0; k < 4; k++) sa.p[k] = g[(ab + k) * np + i];
        have_sa = true;
      }
      g[(base + 4) * np + i] = make_float4(nx4.x, nx4.y, __uint_as_float(0u), 0.f);
      taken = true;
    }
    pool_request(a, i);
    pool_count(a, taken);
  }
  if (!taken) sample_episode<LOAD>(a, i, e.episode, e.s);
  reset_bookkeeping(e.s, e.episode, e.num_steps);
  if (a.obs_needs_acc) {
    // the new episode's first row reads the sensor at the new state (set_state -> mj_forward): from the pool's affine form if the
    // entry carried it, else by running the forward dynamics again
    if (have_sa) { e.acc = sensor_from_affine(sa, e.s.a0, e.s.a1, e.s.a2, e.s.a3); e.flags &= ~FLAG_ACC_STALE; }
    else refresh_sensor<LOAD>(a, e);
  } else {
    e.flags |= FLAG_ACC_STALE;
  }
}

// ---- observation rows ---------------------------------------------------------------
// Each lane produces D floats.  The block's rows are first written to LDS at
// [lane][k] (row-major, i.e. exactly the image of the global span) and then copied
// out with consecutive lanes writing consecutive dwords.
constexpr int OBS_LDS_FLOATS = 64 * QD_MAX_OBS;

template <int NS, int KIND>
__device__ __forceinline__ void obs_to_lds(const float* sv, const float ref[4], float* row, const M3<float>* Rq) {
  float o[QD_MAX_OBS];
  const int n = observe<float, NS, KIND>(sv, ref, o, Rq);
#pragma unroll
  for (int k = 0; k < QD_MAX_OBS; k++)
    if (k < n) row[k] = o[k];
}

template <bool LOAD, int SPEC>
__device__ __forceinline__ void write_obs_row(const KArgs& a, const EnvRegs& e, const float* sv, const M3<float>* Rq,
                                              float* row) {
  constexpr int NS = LOAD ? 33 : 29;
  const int kind = spec_obs<SPEC>(a);
  if (kind == OBS_SIMPLE) {
    float o[6];
    simple_obs<float>(e.s, o);
#pragma unroll
    for (int k = 0; k < 6; k++) row[k] = o[k];
    return;
  }
  if (SPEC == SPEC_RMA) { obs_to_lds<NS, OBS_RPY_PARAMS>(sv, e.ref, row, Rq); return; }
  if (SPEC == SPEC_LSTM) { obs_to_lds<NS, OBS_FULLSTATE>(sv, e.ref, row, Rq); return; }
#define QD_CALL(K) obs_to_lds<NS, K>(sv, e.ref, row, Rq)
  QD_OBS_DISPATCH(kind, QD_CALL)
#undef QD_CALL
}

// copy the wave's staged rows (rows x D floats, contiguous in LDS and in global memory; both 16-byte
// aligned because a wave starts at a multiple of 64 rows): 16 bytes per lane per instruction
// Observation rows are written once and never read back by the env: non-temporal stores keep a long fragment (1024 steps
// x 4096 envs = 369 MB of rows, more than the Infinity Cache) from evicting the state planes and from waiting on HBM write
// acknowledgements at the end of every launch (5.32 -> 4.89 us per step on 1024-step fragments, 4.75 -> 4.65 on 128-step ones).
__device__ __forceinline__ void store_streaming(float4* p, float4 v) {
  typedef float nt_f4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store((nt_f4{v.x, v.y, v.z, v.w}), reinterpret_cast<nt_f4*>(p));
}
__device__ __forceinline__ void flush_obs(const float* tile, float* dst, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int total = rows * D, n4 = total >> 2;
  const float4* t4 = reinterpret_cast<const float4*>(tile);
  float4* d4 = reinterpret_cast<float4*>(dst);
#pragma unroll 3
  for (int j = lane; j < n4; j += 64) store_streaming(d4 + j, t4[j]);
  for (int j = (n4 << 2) + lane; j < total; j += 64) __builtin_nontemporal_store(tile[j], dst + j);
}

// full wavefront, row length known at compile time: all LDS reads are issued before the first store.
// FULL unpredicated rounds keep v[] in registers (a predicated round made the compiler index it dynamically
// and park it in scratch); the partial last round is handled on its own.
template <int D>
__device__ __forceinline__ void flush_obs_static(const float* tile, float* dst) {
  constexpr int N4 = 16 * D, FULL = N4 / 64, TAIL = N4 % 64;
  const int lane = threadIdx.x & 63;
  const float4* t4 = reinterpret_cast<const float4*>(tile) + lane;
  float4* d4 = reinterpret_cast<float4*>(dst) + lane;
  float4 v[FULL > 0 ? FULL : 1];
#pragma unroll
  for (int k = 0; k < FULL; k++) v[k] = t4[64 * k];
  float4 vt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (TAIL > 0 && lane < TAIL) vt = t4[64 * FULL];
#pragma unroll
  for (int k = 0; k < FULL; k++) store_streaming(d4 + 64 * k, v[k]);
  if (TAIL > 0 && lane < TAIL) store_streaming(d4 + 64 * FULL, vt);
}
template <int SPEC> constexpr int spec_obs_dim() { return SPEC == SPEC_RMA ? 22 : SPEC == SPEC_LSTM ? 23 : SPEC == SPEC_SIMPLE ? 6 : 0; }

template <int SPEC>
__device__ __forceinline__ void flush_obs_any(const float* tile, float* dst, int rows, int D) {
  if (!spec_runtime<SPEC>() && rows == 64) flush_obs_static<(spec_obs_dim<SPEC>() > 0 ? spec_obs_dim<SPEC>() : 4)>(tile, dst);
  else flush_obs(tile, dst, rows, D);
}

// Arguments of the step kernels, as they lie in the kernarg segment (same member order as k_step_wide's / k_step_coop's parameters).
// The leading scalars repeat members of the struct: they are all the first round of global loads needs (arena base, plane
// stride, env count, role of the workgroup), and as separate arguments ahead of the struct they are preloaded into SGPRs at
// wave launch (-mllvm -amdgpu-kernarg-preload-count, build.py): those loads leave at once instead of one memory round trip
// later, behind the fetch of the kernarg segment (about as slow as they are: every launch starts with cold caches).
// Measured on one box, alternating runs, config 3: 4.25 -> 4.02 us per step at 4096 envs (k_step_coop), 78.8 -> 70.7 us at 2^20
// envs (k_step<true,256,1>, which also loses its last 36 bytes of scratch); the single-wave 64-thread kernels gained nothing
// in any arrangement tried (config 5 at 8192 envs: 5.42 -> 5.47 ... 5.65 us) and keep their struct-first signature (k_step).
struct StepKernarg {
  float4* g;
  const float* actions;
  int npad, n, main_blocks;
  KArgs a;
  float* obs;
  float* reward;
  uint8_t* trunc;
};
// The KArgs of a step launch, read from the kernarg segment at the point of the call and not before.  A reference to the
// struct parameter itself is lowered to scalar loads at the top of the kernel plus register copies that wait for them there,
// ahead of everything; the empty asm hides the address from the optimiser and keeps the global loads issued so far ahead of
// these scalar loads.  (Tried and slower: the scalar loads first and only the wait deferred -- 4.09 us; every member pinned in
// one batch instead of sunk into the branches that use it -- same at 4096 envs, 3 % slower for config 5.)
__device__ __forceinline__ KArgs step_kargs(float4* g, int npad, int n, int main_blocks) {
  typedef const __attribute__((address_space(4))) char* kptr;
  kptr kp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp)::"memory");
  static_assert(sizeof(KArgs) % 4 == 0 && offsetof(StepKernarg, a) % 4 == 0, "KArgs is copied from the kernarg segment in dwords");
  const __attribute__((address_space(4))) uint32_t* w = (const __attribute__((address_space(4))) uint32_t*)(kp + offsetof(StepKernarg, a));
  constexpr unsigned NW = sizeof(KArgs) / 4;
  uint32_t u[NW];
#pragma unroll
  for (unsigned k = 0; k < NW; k++) u[k] = w[k];   // only the members the kernel reads survive as scalar loads
  KArgs a;
  __builtin_memcpy(&a, u, sizeof(KArgs));
  a.g = g;
  a.npad = npad;
  a.n = n;
  a.main_blocks = main_blocks;
  return a;
}

constexpr int COOP_THREADS = 192;
struct CoopLds {
  float4 app[5][64];     // B -> A: Applied (F, t1) (Tq, t2) and the attitude matrix (9 values)
  double2 ine[4][64];    // C -> A: Inertial (F, Tq, t1, t2)
  float4 nxt[5][64];     // C -> A: the env's reset-pool entry for its current episode counter, as stored (prefetched)
  float4 st[2][6][64];   // A -> B, C: [0] the state after the step, [1] after the in-kernel reset (truncated lanes only):
                         //   (pos, th1) (quat) (vel, th2) (angvel, thd1) (act) (thd2, accelerometer)
  uint32_t flag[64];     // A -> B, C: the lane was reset (its observation row is the new episode's first)
  float tile[64 * 24];   // the group's observation rows, row-major like the global span
};
// observation slots whose value depends on the attitude matrix / yaw (wave C); the rest are wave B's
template <int SPEC> constexpr unsigned coop_frame_slots() { return 0x1E7u; }  // e_l 0..2, heading 5, v_l 6..8 (RPY_PARAMS and FULLSTATE)

// Workgroup barrier for an LDS hand-over: waits for this wave's LDS operations only.  __syncthreads() also drains the wave's
// global stores and atomics (vmcnt(0)) -- a ~1300-cycle round trip whenever wave A has just consumed a pool entry.
__device__ __forceinline__ void coop_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void coop_put_state(CoopLds& L, int set, int lane, const State<float>& s, V3<float> acc) {
  L.st[set][0][lane] = make_float4(s.px, s.py, s.pz, s.th1);
  L.st[set][1][lane] = make_float4(s.qw, s.qx, s.qy, s.qz);
  L.st[set][2][lane] = make_float4(s.vx, s.vy, s.vz, s.th2);
  L.st[set][3][lane] = make_float4(s.wx, s.wy, s.wz, s.thd1);
  L.st[set][4][lane] = make_float4(s.a0, s.a1, s.a2, s.a3);
  L.st[set][5][lane] = make_float4(s.thd2, acc.x, acc.y, acc.z);
}
__device__ __forceinline__ void coop_get_state(const CoopLds& L, int set, int lane, State<float>& s, V3<float>& acc) {
  const float4 p = L.st[set][0][lane], q = L.st[set][1][lane], v = L.st[set][2][lane], w = L.st[set][3][lane];
  const float4 c = L.st[set][4][lane], x = L.st[set][5][lane];
  s.px = p.x; s.py = p.y; s.pz = p.z; s.th1 = p.w;
  s.qw = q.x; s.qx = q.y; s.qy = q.z; s.qz = q.w;
  s.vx = v.x; s.vy = v.y; s.vz = v.z; s.th2 = v.w;
  s.wx = w.x; s.wy = w.y; s.wz = w.z; s.thd1 = w.w;
  s.a0 = c.x; s.a1 = c.y; s.a2 = c.z; s.a3 = c.w;
  s.thd2 = x.x; acc = mk<float>(x.y, x.z, x.w);
}

// one wave's share of the observation row: FRAME = the slots that need the attitude matrix / yaw (wave C), else the rest (wave B).
// Both instantiate the full observe<>() of the variant; what a wave does not store is dead code for it.
template <int SPEC, bool FRAME>
__device__ __forceinline__ void coop_obs_part(const KArgs& a, int i, int lane, const EnvRegs& e, CoopLds& L) {
  constexpr int D = spec_obs_dim<SPEC>();
  constexpr int KIND = SPEC == SPEC_RMA ? (int)OBS_RPY_PARAMS : (int)OBS_FULLSTATE;
  constexpr unsigned MASK = coop_frame_slots<SPEC>();
  const bool rst = L.flag[lane] != 0u;
  State<float> s;
  V3<float> sacc;
  coop_get_state(L, rst ? 1 : 0, lane, s, sacc);
  float ref[4] = {e.ref[0], e.ref[1], e.ref[2], e.ref[3]};
  if (rst && a.ref_mode != QD_REF_STATIC) moving_reference(a, i, 0, ref);
  float sv[33], o[QD_MAX_OBS];
  M3<float> Rq;
  drone_state<float, true>(s, sacc, ref, e.par, sv, &Rq);
  observe<float, 33, KIND>(sv, ref, o, &Rq);
  // through the LDS tile: each wave storing its slots straight to the rows in global memory (4-byte stores, 88-byte lane
  // stride) would save the third barrier and the flush, and was measured 15 % slower (4.85 against 4.24 us per step)
  float* row = L.tile + lane * D;
#pragma unroll
  for (int k = 0; k < D; k++)
    if ((((MASK >> k) & 1u) != 0u) == FRAME) row[k] = o[k];
}

// ---- analytic PID cascade (models/Analytic/*.py driven as attitude_test.py:36-47) ----
__device__ __forceinline__ void load_pid(const KArgs& a, int i, PidState<float>& c) {
  const float4 c0 = a.g[G_C0 * a.npad + i], c1 = a.g[G_C1 * a.npad + i], c2 = a.g[G_C2 * a.npad + i], c3 = a.g[G_C3 * a.npad + i];
  c.pos_i[0] = c0.x; c.pos_i[1] = c0.y; c.pos_i[2] = c0.z; c.first = __float_as_uint(c0.w);
  c.pos_prev[0] = c1.x; c.pos_prev[1] = c1.y; c.pos_prev[2] = c1.z;
  c.att_i[0] = c2.x; c.att_i[1] = c2.y; c.att_i[2] = c2.z;
  c.att_prev[0] = c3.x; c.att_prev[1] = c3.y; c.att_prev[2] = c3.z;
}
__device__ __forceinline__ void store_pid(const KArgs& a, int i, const PidState<float>& c) {
  a.g[G_C0 * a.npad + i] = make_float4(c.pos_i[0], c.pos_i[1], c.pos_i[2], __uint_as_float(c.first));
  a.g[G_C1 * a.npad + i] = make_float4(c.pos_prev[0], c.pos_prev[1], c.pos_prev[2], 0.f);
  a.g[G_C2 * a.npad + i] = make_float4(c.att_i[0], c.att_i[1], c.att_i[2], 0.f);
  a.g[G_C3 * a.npad + i] = make_float4(c.att_prev[0], c.att_prev[1], c.att_prev[2], 0.f);
}
// the controller's inputs are entries 0:6 of the drone state vector (get_drone_states: xyz, rpy of the normalised quaternion)
__device__ __forceinline__ float4 pid_env_action(PidState<float>& c, const EnvRegs& e) {
  const float qn = frsq(e.s.qw * e.s.qw + e.s.qx * e.s.qx + e.s.qy * e.s.qy + e.s.qz * e.s.qz);
  const float xyz[3] = {e.s.px, e.s.py, e.s.pz};
  float rpy[3], act[4];
  quat2rpy(e.s.qw * qn, e.s.qx * qn, e.s.qy * qn, e.s.qz * qn, &rpy[0], &rpy[1], &rpy[2]);
  pid_action(c, e.ref, xyz, rpy, pid_mass(e.par), pid_motor_force(e.par), act);
  return make_float4(act[0], act[1], act[2], act[3]);
}

// ---- pieces the persistent role kernels share (qd_rollout_coop.hip, qd_rollout_fused.hip) ----
// The part of the Euler step that reads the action: the reference's ctrl map (BaseDroneEnv.step: 0.1 + 0.9 u for the training
// configurations), MuJoCo's clamp to the ctrl range, and the activation filter.  One definition: the persistent kernels apply it in
// several waves (to the solver's state, to the thrust wave's copy, to rows that carry the activations) and must agree bit for bit.
template <int SPEC>
__device__ __forceinline__ void rc_filter(const KArgs& a, const Model<float>& M, State<float>& s, float4 u) {
  float c0 = u.x, c1 = u.y, c2 = u.z, c3 = u.w;
  if (spec_ctrl<SPEC>(a) == QD_CTRL_AFFINE) { c0 = 0.1f + 0.9f * c0; c1 = 0.1f + 0.9f * c1; c2 = 0.1f + 0.9f * c2; c3 = 0.1f + 0.9f * c3; }
  integrate_act(M, s, qclamp(c0, 0.f, 1.f), qclamp(c1, 0.f, 1.f), qclamp(c2, 0.f, 1.f), qclamp(c3, 0.f, 1.f), a.h);
}

// where the three accelerometer values sit in the observation row of a variant that carries them (-1: it does not)
__device__ __forceinline__ int rc_acc_slot(int kind) {
  switch (kind) {
    case OBS_RAW: return 16;
    case OBS_FULLSTATE: case OBS_PRY_ACC: case OBS_PRY_ACC_NOPEND: return 12;
    case OBS_FULLSTATE_ZVEC: return 13;
    case OBS_PRY_ACC_PARAMS: return 14;
    default: return -1;
  }
}
// the env's reference at episode step k (static: `base`, fetched once)
__device__ __forceinline__ void rc_ref(const KArgs& a, int i, int k, const float base[4], float ref[4]) {
  if (a.ref_mode != QD_REF_STATIC) moving_reference(a, i, k, ref);
  else { ref[0] = base[0]; ref[1] = base[1]; ref[2] = base[2]; ref[3] = base[3]; }
}
// "this value exists here": an empty asm that reads it (the volatile asms keep their order, the barrier among them)
__device__ __forceinline__ void rc_pin(double x, double y, double z) { asm volatile("" ::"v"(x), "v"(y), "v"(z)); }
__device__ __forceinline__ void rc_pin(const V3<double>& v) { rc_pin(v.x, v.y, v.z); }

// the accelerometer reading of a step from its factor and right-hand side (what k_step_coop's phase 3 computes)
__device__ __forceinline__ V3<float> rc_sensor(const Factor<double>& f, const Rhs<double>& r, const M3<float>& R, V3<float> w0) {
  Accel<float> ex;
  V3<double> a0ex;
  finish_accel<false, true>(f, r, &a0ex, &ex.ang, &ex.thdd1, &ex.thdd2);
  const float g = float(Const::gravity);
  return accelerometer(cvt<float>(a0ex), ex.ang, mk<float>(g * R.m20, g * R.m21, g * R.m22),
                       mk<float>(w0.x * w0.z, w0.y * w0.z, -(w0.x * w0.x + w0.y * w0.y)));
}


// qd_rollout_coop.hip: T steps of the load model (SPEC_RMA, SPEC_LSTM or SPEC_GENERIC_FS1: one substep per step) in ONE launch,
// four wavefronts per 64 envs.  `k` as qd_step would pass it (main_blocks is set by the launcher).
// `pid`: the analytic PID cascade (qd_pid.h) is the action source (actions = nullptr, actions_out [T,N,4] nullable), else `actions` [T,N,4].
hipError_t launch_rollout_coop(const KArgs& k, int spec, int T, const float* actions, float* obs, float* reward, uint8_t* trunc, hipStream_t stream,
                               bool pid = false, float* actions_out = nullptr);

// qd_rollout_lat.hip: the same fragment for batches of at most 256 workgroups (every workgroup has a CU to itself), SPEC_RMA or
// SPEC_GENERIC_FS1 with an observation variant that does not read the accelerometer; hipErrorInvalidValue otherwise
hipError_t launch_rollout_lat(const KArgs& k, int spec, int T, const float* actions, float* obs, float* reward, uint8_t* trunc, hipStream_t stream);
// SimpleDrone's fragments at small batches: a physics wave and an epilogue wave per 64 envs (qd_rollout_coop.hip)
hipError_t launch_rollout_pair(const KArgs& k, int T, const float* actions, float* obs, float* reward, uint8_t* trunc, hipStream_t stream);
// qd_step_floor.hip: the floor-contact constants of the current parameter set (after every k_params of a floor-contact env), and
// one env step of a floor-contact configuration (SPEC_FLOOR), any batch size
hipError_t launch_floor_consts(const KArgs& k, bool load, hipStream_t stream);
hipError_t launch_step_floor(const KArgs& k, bool load, const float* actions, float* obs, float* reward, uint8_t* trunc, hipStream_t stream);

}  // namespace qd
