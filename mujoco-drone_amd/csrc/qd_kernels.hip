// qd_kernels.hip -- gfx950 kernels and the C ABI (include/qd.h) of the vectorised
// quadrotor(+load) environment.  One env per lane, 64-wide wavefronts.
//
// HBM layout ("arena", caller-owned): NUM_GROUPS planes of float4[npad], i.e. every
// lane moves its state with 16-byte loads/stores and a wavefront touches 1 KiB of
// consecutive memory per instruction (the widest coalesced access on CDNA4), followed by
// 6 planes of double[npad] holding the raw float64 model parameters:
//   POS  (px,py,pz,th1)   QUAT (qw,qx,qy,qz)   VEL (vx,vy,vz,th2)   ANG (wx,wy,wz,thd1)
//   ACT  (a0..a3)         AUX  (thd2, num_steps:i32, episode:u32, flags:u32)
//   ACC  (accelerometer xyz, -)                 M0..M6 derived model constants (qd_model.h)
//   P0   (mass, arm_len, motor_force, motor_tau) P1 (pendulum_len, weight_mass, -, -)
//   REF  (x,y,z,yaw) per-env reference (only read when per_env_reference is set)
//   NX0..NX4 / NY0..NY4 the pre-sampled initial states of the env's next two episodes (even / odd episode counter):
//        (pos,th1) (quat) (vel,th2) (angvel,thd1) (thd2, episode tag:u32, valid:u32, -), see "reset pool" below
// behind the planes: RAW x 6 float64 planes, then one uint32 per 64 envs: refill requests of the reset pool
// Observations are produced row-major [N,D] (what the policy network consumes); a
// wavefront's 64 rows are one contiguous 64*D*4-byte span, so rows are staged through LDS
// and written back with full-width coalesced stores.
#include <vector>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/qd.h"
#include "qd_contact.h"
#include "qd_dynamics.h"
#include "qd_env_device.h"
#include "qd_math.h"
#include "qd_model.h"
#include "qd_obsrew.h"
#include "qd_pid.h"
#include "qd_policy.h"
#include "qd_policy_static.h"
#include "qd_rng.h"
#include "qd_stats.h"

namespace qd {
#ifdef QD_STAMPS
// diagnostic build only: per-wave cycle stamps of the step kernel's phases (cdna_hip_programming.md, In-kernel stamps)
__device__ unsigned long long qd_stamps[64 * 8];
__device__ unsigned long long qd_rstamps[64 * 2];
#define QD_STAMP(k)                                                                                   \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    unsigned long long t_;                                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) qd_stamps[blockIdx.x * 8 + (k)] = t_;           \
  } while (0)
// cooperative step: 16 stamps per (workgroup, wave)
__device__ unsigned long long qd_cstamps[64 * 3 * 16];
#define QD_CSTAMP(k)                                                                                        \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    unsigned long long t_;                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                            \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) qd_cstamps[(blockIdx.x * 3 + role) * 16 + (k)] = t_;   \
  } while (0)
#else
#define QD_STAMP(k)
#define QD_CSTAMP(k)
#endif

// ---- one full env step for the lane's env (everything after the state is in registers) ----
template <bool LOAD, int SPEC, bool POOL = true>
__device__ __forceinline__ void env_step(const KArgs& a, int i, EnvRegs& e, float4 action, float* obs_row, float* rew,
                                         uint8_t* trunc) {
  constexpr int NS = LOAD ? 33 : 29;
  float c0 = action.x, c1 = action.y, c2 = action.z, c3 = action.w;
  if (spec_ctrl<SPEC>(a) == QD_CTRL_AFFINE) { c0 = 0.1f + 0.9f * c0; c1 = 0.1f + 0.9f * c1; c2 = 0.1f + 0.9f * c2; c3 = 0.1f + 0.9f * c3; }
  c0 = qclamp(c0, 0.f, 1.f); c1 = qclamp(c1, 0.f, 1.f); c2 = qclamp(c2, 0.f, 1.f); c3 = qclamp(c3, 0.f, 1.f);
  static_assert(SPEC != SPEC_FLOOR, "floor-contact configurations are stepped by k_step_floor (qd_step_floor.hip)");
  if (spec_frame_skip<SPEC>() == 1) {
    e.acc = substep<float, LOAD>(e.M, e.s, c0, c1, c2, c3, a.h);
  } else if (spec_frame_skip<SPEC>() == 2) {
    e.acc = substep<float, LOAD>(e.M, e.s, c0, c1, c2, c3, a.h);
    e.acc = substep<float, LOAD>(e.M, e.s, c0, c1, c2, c3, a.h);
  } else {
    for (int k = 0; k < a.frame_skip; k++) e.acc = substep<float, LOAD>(e.M, e.s, c0, c1, c2, c3, a.h);
  }
  QD_STAMP(2);
  e.flags &= ~FLAG_ACC_STALE;
  e.num_steps += 1;
  float sv[33];
  const float act4[4] = {action.x, action.y, action.z, action.w};
  M3<float> Rq;
  bool tr;
  float r;
  const int term_kind = spec_term<SPEC>(a);
  if (term_kind == QD_TERM_SIMPLE) {
    // SimpleDrone.step: terminated = |pos - ref| > 0.5, reward = 0.1 - |pos - ref| (SimpleDrone.py:57-60)
    const float dx = e.s.px - e.ref[0], dy = e.s.py - e.ref[1], dz = e.s.pz - e.ref[2];
    const float d = qsqrt(dx * dx + dy * dy + dz * dz);
    tr = d > 0.5f;
    r = 0.1f - d;
  } else {
    drone_state<float, LOAD>(e.s, e.acc, e.ref, e.par, sv, &Rq);
    tr = truncated<float>(sv, e.ref, e.num_steps, a.max_distance, a.max_steps);
    r = reward<float>(spec_reward<SPEC>(a), sv, act4, e.num_steps, e.ref, a.max_distance, &Rq);
  }
  if (a.auto_reset && tr) {
    reset_in_step<LOAD, POOL>(a, i, e);
    if (a.ref_mode != QD_REF_STATIC) moving_reference(a, i, 0, e.ref);
    if (term_kind != QD_TERM_SIMPLE) drone_state<float, LOAD>(e.s, e.acc, e.ref, e.par, sv, &Rq);
  }
  *rew = r;
  *trunc = tr ? 1 : 0;
  QD_STAMP(3);
  write_obs_row<LOAD, SPEC>(a, e, sv, &Rq, obs_row);
  (void)NS;
}

// The 256-thread variant (>= 65536 envs, the chip is full) is capped at 256 registers so that TWO waves share a
// SIMD and hide each other's memory latency; with the substep count known at compile time the specialised
// instantiations fit with a few dwords of scratch.  The 64-thread variant (small batches: one wave per SIMD
// anyway) keeps the whole register file.
// DEFER: the kernel has the StepKernarg signature (k_step_wide); else `a_in` is the kernel's own struct parameter, read at the top
template <bool LOAD, int BLOCK, int SPEC, bool DEFER>
__device__ __forceinline__ void step_body(float4* g_pre, const float* __restrict__ actions, int npad_pre, int n_pre, int main_blocks_pre,
                                          const KArgs& a_in, float* __restrict__ obs, float* __restrict__ reward,
                                          uint8_t* __restrict__ trunc) {
  __shared__ float tile[(BLOCK / 64) * OBS_LDS_FLOATS];
  if (BLOCK == 64 && (int)blockIdx.x >= a_in.main_blocks) {  // sampler workgroup (see "reset pool"); 64-thread launches only
    sampler_wave<LOAD>(a_in, ((int)blockIdx.x - a_in.main_blocks) * BLOCK + threadIdx.x);
    return;
  }
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* wtile = tile + wave * OBS_LDS_FLOATS;
  const int wave_base = i - lane;  // first env of this wavefront
#ifdef QD_STAMPS
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) qd_rstamps[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
#endif
  QD_STAMP(0);
  EnvRegs e;
  float4 action = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (DEFER) {
    // lanes past the batch fetch the last env's planes: an unconditional fetch keeps the loads, the argument fetch and the
    // first arithmetic in one basic block, in that order
    const int il = i < n_pre ? i : n_pre - 1;
    load_env_planes<LOAD, false, true>(g_pre, npad_pre, il, e);
    action = reinterpret_cast<const float4*>(actions)[il];
  }
  const KArgs a = DEFER ? step_kargs(g_pre, npad_pre, n_pre, main_blocks_pre) : a_in;
  if (i < a.n) {
    if constexpr (DEFER) {
      load_env_ref(a, i, e);
    } else {
      load_env<LOAD, false, false>(a, i, e);
      action = reinterpret_cast<const float4*>(actions)[i];
    }
#ifdef QD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    QD_STAMP(1);
#endif
    float r;
    uint8_t t;
    env_step<LOAD, SPEC, BLOCK == 64>(a, i, e, action, wtile + lane * a.D, &r, &t);
    QD_STAMP(4);
    store_env(a, i, e);
    __builtin_nontemporal_store(r, reward + i);
    __builtin_nontemporal_store(t, trunc + i);
  }
  // wave-local staging: the LDS tile is private to the wavefront, so no block barrier is needed
  __builtin_amdgcn_wave_barrier();
  QD_STAMP(5);
  if (wave_base < a.n) {
    const int rows = min(64, a.n - wave_base);
    flush_obs_any<SPEC>(wtile, obs + (size_t)wave_base * a.D, rows, a.D);
  }
  QD_STAMP(6);
#ifdef QD_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  QD_STAMP(7);
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) qd_rstamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// 64-thread workgroups (one wavefront per SIMD while the batch fits the chip once)
template <bool LOAD, int BLOCK, int SPEC>
__global__ __launch_bounds__(BLOCK, 1) void k_step(KArgs a, const float* __restrict__ actions, float* __restrict__ obs,
                                                   float* __restrict__ reward, uint8_t* __restrict__ trunc) {
  static_assert(BLOCK == 64, "k_step: 64-thread launches; 256-thread launches are k_step_wide");
  step_body<LOAD, BLOCK, SPEC, false>(a.g, actions, a.npad, a.n, a.main_blocks, a, obs, reward, trunc);
}
// 256-thread workgroups, two wavefronts per SIMD (see above), arguments as in StepKernarg
template <bool LOAD, int BLOCK, int SPEC>
__global__ __launch_bounds__(BLOCK, (!spec_runtime<SPEC>() ? 2 : 1)) void k_step_wide(float4* g_pre, const float* __restrict__ actions, int npad_pre,
                                                                                    int n_pre, int main_blocks_pre, KArgs a_in,
                                                                                    float* __restrict__ obs, float* __restrict__ reward,
                                                                                    uint8_t* __restrict__ trunc) {
  static_assert(BLOCK == 256, "k_step_wide: 256-thread launches");
  step_body<LOAD, BLOCK, SPEC, true>(g_pre, actions, npad_pre, n_pre, main_blocks_pre, a_in, obs, reward, trunc);
}

// ---- cooperative step: THREE wavefronts per 64 envs ------------------------------------------------------------
// At 4096 envs the single-wave step is a latency chain: one wavefront alone on its SIMD issues one vector instruction
// per 4 cycles, so its ~1300 instructions ARE the kernel time (3 of the 5 us period), while 94 % of the chip idles.
// The load model's forward dynamics has three mutually independent parts of similar length -- the applied wrench
// (thrust + drag on three bodies, float32), the inertial wrench (gravity + velocity products, float64) and the
// mass-matrix factorisation (float64), qd_dynamics.h -- and so has the epilogue (reward / state stores, attitude half
// of the observation row, position half of it).  k_step_coop runs them in three wavefronts of one workgroup (three
// SIMDs of a CU), lane l of each wave working on env l of the group, exchanging 16 + 16 + 24 dwords per env through
// LDS at three workgroup barriers:
//   wave A: mass_factor                 | reduce_rhs, finish_accel(im), integrate, truncation, (reset) | finish_accel(ex),
//                                       |                                                               accelerometer, reward, stores
//   wave B: attitude, applied_wrench    | -                                                            | roll / pitch / hinge / param slots
//   wave C: inertial_wrench, pool entry | -                                                            | frame-dependent slots (e_l, heading, v_l)
// Same arithmetic as the single-wave step (the same functions, composed across waves instead of in one lane).
template <int SPEC>
__global__ __launch_bounds__(COOP_THREADS) void k_step_coop(float4* g_pre, const float* __restrict__ actions, int npad_pre, int n_pre,
                                                            int main_blocks_pre, KArgs a_in, float* __restrict__ obs,
                                                            float* __restrict__ reward_out, uint8_t* __restrict__ trunc_out) {
  // arguments: see step_kargs
  static_assert(SPEC == SPEC_RMA, "cooperative step: LocalFrameRPYParamsEnv + distance_energy_reward on the load model");
  constexpr int D = spec_obs_dim<SPEC>();
  constexpr int KIND = SPEC == SPEC_RMA ? (int)OBS_RPY_PARAMS : (int)OBS_FULLSTATE;
  __shared__ CoopLds L;
  if ((int)blockIdx.x >= main_blocks_pre) {  // sampler workgroup (see "reset pool")
    const KArgs a = step_kargs(g_pre, npad_pre, n_pre, main_blocks_pre);
    sampler_wave<true>(a, ((int)blockIdx.x - main_blocks_pre) * COOP_THREADS + threadIdx.x);
    return;
  }
  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < n_pre;
  QD_CSTAMP(0);
  // every wave fetches all of its env's planes, not just the ones its role reads: fetching per role, in order of need, was
  // measured 6 % SLOWER (4.48 against 4.22 us per step, same box, alternating runs) although it moves a third fewer bytes;
  // non-live lanes only keep the barriers company
  // (they fetch the last env's planes: an unconditional fetch keeps the loads, the argument fetch and the first arithmetic in
  // one basic block, in that order)
  EnvRegs e;
  const int il = live ? i : n_pre - 1;
  load_env_planes<true, false, false>(g_pre, npad_pre, il, e);
  const float4 action = reinterpret_cast<const float4*>(actions)[il];
  const KArgs a = step_kargs(g_pre, npad_pre, n_pre, main_blocks_pre);
  if (live) load_env_ref(a, i, e);
  Factor<double> f;
#ifdef QD_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  QD_CSTAMP(1);
  // ---------------------------------------------------------------- phase 1: three independent parts
  if (live) {
    const Tether<float> tg = tether_geometry(e.s.th1, e.s.th2);
    if (role == 0) {
      f = mass_factor(e.M, tg, a.h);
      // the part of the Euler step that does not wait for the accelerations: ctrl map and the activation filter
      const float c0 = qclamp(0.1f + 0.9f * action.x, 0.f, 1.f), c1 = qclamp(0.1f + 0.9f * action.y, 0.f, 1.f);
      const float c2 = qclamp(0.1f + 0.9f * action.z, 0.f, 1.f), c3 = qclamp(0.1f + 0.9f * action.w, 0.f, 1.f);
      integrate_act(e.M, e.s, c0, c1, c2, c3, a.h);
    } else if (role == 1) {
      const Att<float> at = attitude(e.s);
      const Applied<float> ap = applied_wrench(e.M, e.s, at, tg);
      L.app[0][lane] = make_float4(ap.F.x, ap.F.y, ap.F.z, ap.t1);
      L.app[1][lane] = make_float4(ap.Tq.x, ap.Tq.y, ap.Tq.z, ap.t2);
      L.app[2][lane] = make_float4(at.R.m00, at.R.m01, at.R.m02, at.R.m10);
      L.app[3][lane] = make_float4(at.R.m11, at.R.m12, at.R.m20, at.R.m21);
      L.app[4][lane] = make_float4(at.R.m22, 0.f, 0.f, 0.f);
    } else {
      // the pool entry the env would take if it truncates in this step: requested first, stored last
      // Only a lane that CAN end its episode in this step fetches the entry: the origin moves |v| h per step, so an env
      // further than that (x2, + 5 cm) inside the bound and not on its last step cannot truncate; should one do so all the
      // same (non-finite state), wave A finds no valid entry and samples inline -- slower, same result.  Keeps the pool's
      // five planes out of the traffic of the ~99 % of lanes that do not need them.
      float4 nx[5];
#pragma unroll
      for (int k = 0; k < 5; k++) nx[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.use_pool && a.auto_reset) {
        const float dx = e.s.px - e.ref[0], dy = e.s.py - e.ref[1], dz = e.s.pz - e.ref[2];
        const float reach = qsqrt(dx * dx + dy * dy + dz * dz) + 2.f * a.h * (float)a.frame_skip * qsqrt(e.s.vx * e.s.vx + e.s.vy * e.s.vy + e.s.vz * e.s.vz) + 0.05f;
        if (!(reach <= a.max_distance) || e.num_steps + 1 >= a.max_steps) {
          const int base = pool_slot(e.episode);
#pragma unroll
          for (int k = 0; k < 5; k++) nx[k] = a.g[(base + k) * a.npad + i];
        }
      }
      V3<float> gt, w;
      gravity_body(e.s, &gt, &w);
      const Inertial<double> in = inertial_wrench(e.M, e.s, gt, w, tg);
      L.ine[0][lane] = make_double2(in.F.x, in.F.y);
      L.ine[1][lane] = make_double2(in.F.z, in.Tq.x);
      L.ine[2][lane] = make_double2(in.Tq.y, in.Tq.z);
      L.ine[3][lane] = make_double2(in.t1, in.t2);
#pragma unroll
      for (int k = 0; k < 5; k++) L.nxt[k][lane] = nx[k];
    }
  }
  // EVERY plane waited for before the first barrier, whether the wave's role read it or not.  Left to itself the compiler
  // waits plane by plane as values are needed and never for the planes a role does not read -- until the wave reuses one of
  // their registers, by which time (phase 3) the state stores are in flight, and the only way to wait for that old load is
  // to wait for the stores behind it too (s_waitcnt vmcnt(3) in the middle of the observation arithmetic: 4.30 us per step
  // against 4.00).  (Until this was understood the kernel owed the same effect to an accident: FLAT atomics on the refill
  // counters put the compiler's wait insertion into its conservative mode, one vmcnt(0) at the top -- 4.02 us.)
  asm volatile("" ::"v"(e.s.px), "v"(e.s.qw), "v"(e.s.vx), "v"(e.s.wx), "v"(e.s.a0), "v"(e.s.thd2), "v"(e.M.m0), "v"(e.M.I0z),
               "v"(e.M.inv_tau), "v"(e.M.klin0), "v"(e.M.qly0), "v"(e.M.qaz0), "v"(e.M.qla2), "v"(e.par[0]), "v"(e.par[4]),
               "v"(action.x));
  QD_CSTAMP(2);
  coop_barrier();
  QD_CSTAMP(3);
  // ---------------------------------------------------------------- phase 2 (wave A): solve, integrate, truncation, reset
  Rhs<double> r;
  M3<float> R;
  V3<float> w0 = mk<float>(0.f, 0.f, 0.f), acc = mk<float>(0.f, 0.f, 0.f);
  State<float> post;   // the state the reward is computed from (after the step, before a reset)
  int steps_post = 0;
  bool tr = false;
  if (role == 0 && live) {
    Applied<float> ap;
    {
      const float4 x0 = L.app[0][lane], x1 = L.app[1][lane], x2 = L.app[2][lane], x3 = L.app[3][lane], x4 = L.app[4][lane];
      ap.F = mk<float>(x0.x, x0.y, x0.z); ap.t1 = x0.w;
      ap.Tq = mk<float>(x1.x, x1.y, x1.z); ap.t2 = x1.w;
      R.m00 = x2.x; R.m01 = x2.y; R.m02 = x2.z; R.m10 = x2.w; R.m11 = x3.x; R.m12 = x3.y; R.m20 = x3.z; R.m21 = x3.w; R.m22 = x4.x;
    }
    Inertial<double> in;
    {
      const double2 y0 = L.ine[0][lane], y1 = L.ine[1][lane], y2 = L.ine[2][lane], y3 = L.ine[3][lane];
      in.F = mk<double>(y0.x, y0.y, y1.x); in.Tq = mk<double>(y1.y, y2.x, y2.y); in.t1 = y3.x; in.t2 = y3.y;
    }
    r = reduce_rhs(f, ap, in);
    Accel<float> im;
    V3<double> a0im;
    finish_accel<true>(f, r, &a0im, &im.ang, &im.thdd1, &im.thdd2);
    im.lin = mul(R, cvt<float>(a0im));
    w0 = mk<float>(e.s.wx, e.s.wy, e.s.wz);
    integrate_motion<float, true>(e.s, im, a.h);
    e.flags &= ~FLAG_ACC_STALE;
    e.num_steps += 1;
    steps_post = e.num_steps;
    post = e.s;
    {  // default_termination_fcn on the position alone (the same test truncated() makes on the state vector)
      const float dx = e.s.px - e.ref[0], dy = e.s.py - e.ref[1], dz = e.s.pz - e.ref[2];
      tr = !(qsqrt(dx * dx + dy * dy + dz * dz) <= a.max_distance) || e.num_steps >= a.max_steps;
    }
    coop_put_state(L, 0, lane, e.s, acc);
    const bool rst = a.auto_reset && tr;
    bool taken = false;
    float4 nx4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rst) {
      State<float> ns;   // the new episode's state; the activations carry over (reset_bookkeeping)
      if (a.use_pool) {
        nx4 = L.nxt[4][lane];
        taken = __float_as_uint(nx4.z) != 0u && __float_as_uint(nx4.y) == e.episode;
      }
      if (taken) {
        const float4 p = L.nxt[0][lane], q = L.nxt[1][lane], v = L.nxt[2][lane], w = L.nxt[3][lane];
        ns.px = p.x; ns.py = p.y; ns.pz = p.z; ns.th1 = p.w;
        ns.qw = q.x; ns.qx = q.y; ns.qy = q.z; ns.qz = q.w;
        ns.vx = v.x; ns.vy = v.y; ns.vz = v.z; ns.th2 = v.w;
        ns.wx = w.x; ns.wy = w.y; ns.wz = w.z; ns.thd1 = w.w;
        ns.thd2 = nx4.x;
      } else {
        sample_episode<true>(a, i, e.episode, ns);
      }
      ns.a0 = e.s.a0; ns.a1 = e.s.a1; ns.a2 = e.s.a2; ns.a3 = e.s.a3;
      e.s = ns;
      const uint32_t consumed = e.episode;
      reset_bookkeeping(e.s, e.episode, e.num_steps);
      e.flags |= FLAG_ACC_STALE;   // the row does not carry the sensor: recomputed by the next step, or by a getter that runs first
      coop_put_state(L, 1, lane, e.s, e.acc);
      // global side effects last: nothing in this phase waits behind them (memory operations retire in order)
      if (a.use_pool) {
        if (taken) a.g[(pool_slot(consumed) + 4) * a.npad + i] = make_float4(nx4.x, nx4.y, __uint_as_float(0u), 0.f);
        pool_request(a, i);
        pool_count(a, taken);
      }
    }
    L.flag[lane] = rst ? 1u : 0u;
  }
  QD_CSTAMP(4);
  coop_barrier();
  QD_CSTAMP(5);
  // ---------------------------------------------------------------- phase 3: epilogue, split three ways
  if (live) {
    if (role == 0) {
      store_env_state(a, i, e);   // positions / velocities / activations / counters: on their way while the rest is computed
      {
        Accel<float> ex;
        V3<double> a0ex;
        finish_accel<false>(f, r, &a0ex, &ex.ang, &ex.thdd1, &ex.thdd2);
        const float g = float(Const::gravity);
        acc = accelerometer(cvt<float>(a0ex), ex.ang, mk<float>(g * R.m20, g * R.m21, g * R.m22),
                            mk<float>(w0.x * w0.z, w0.y * w0.z, -(w0.x * w0.x + w0.y * w0.y)));
        e.acc = acc;   // stored even when a reset marked it stale: it is this step's reading (single-wave step: same)
      }
      a.g[G_ACC * a.npad + i] = make_float4(e.acc.x, e.acc.y, e.acc.z, 0.f);
      float sv[33];
      M3<float> Rq;
      const float act4[4] = {action.x, action.y, action.z, action.w};
      drone_state<float, true>(post, acc, e.ref, e.par, sv, &Rq);
      const float rw = reward<float>(spec_reward<SPEC>(a), sv, act4, steps_post, e.ref, a.max_distance, &Rq);
      __builtin_nontemporal_store(rw, reward_out + i);
      __builtin_nontemporal_store((uint8_t)(tr ? 1 : 0), trunc_out + i);
    } else if (role == 1) {
      coop_obs_part<SPEC, false>(a, i, lane, e, L);
    } else {
      coop_obs_part<SPEC, true>(a, i, lane, e, L);
    }
  }
  QD_CSTAMP(6);
  coop_barrier();
  QD_CSTAMP(7);
  {  // the group's rows are one contiguous span: 16-byte streaming stores by all three waves
    const int base_env = blockIdx.x * 64;
    const int rows = min(64, a.n - base_env);
    const int total = rows * D, n4 = total >> 2;
    const float4* t4 = reinterpret_cast<const float4*>(L.tile);
    float4* d4 = reinterpret_cast<float4*>(obs + (size_t)base_env * D);
    // two rounds cover the tile (64 rows x <= 24 floats = 384 chunks = 2 x 192 threads): both LDS reads in flight before the
    // first store, no loop around one read-wait-store at a time
    static_assert(sizeof(L.tile) / 16 <= 2 * COOP_THREADS, "flush: two rounds of 16-byte chunks");
    const int j0 = threadIdx.x, j1 = threadIdx.x + COOP_THREADS;
    const float4 c0 = t4[j0], c1 = t4[j1];   // inside the tile whatever n4 is; only the stores are conditional
    asm volatile("" ::"v"(c0.x), "v"(c1.x));  // (both reads ahead of the first branch: the compiler sinks them into the branches)
    if (j0 < n4) store_streaming(d4 + j0, c0);
    if (j1 < n4) store_streaming(d4 + j1, c1);
    for (int j = (n4 << 2) + threadIdx.x; j < total; j += COOP_THREADS) __builtin_nontemporal_store(L.tile[j], obs + (size_t)base_env * D + j);
  }
  QD_CSTAMP(8);
#ifdef QD_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  QD_CSTAMP(9);
#endif
}

// T steps per launch, state in registers between steps
template <bool LOAD, int BLOCK, int SPEC>
__global__ __launch_bounds__(BLOCK) void k_rollout(KArgs a, int T, const float* __restrict__ actions,
                                                   float* __restrict__ obs, float* __restrict__ reward,
                                                   uint8_t* __restrict__ trunc) {
  __shared__ float tile[(BLOCK / 64) * OBS_LDS_FLOATS];
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* wtile = tile + wave * OBS_LDS_FLOATS;
  const int wave_base = i - lane;
  const bool live = i < a.n;
  EnvRegs e;
  if (live) load_env<LOAD, false>(a, i, e);
  float4 next_action = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && T > 0) next_action = reinterpret_cast<const float4*>(actions)[i];
  for (int t = 0; t < T; t++) {
    if (live) {
      const float4 action = next_action;
      // software pipelining: the next step's action is in flight while this step computes
      if (t + 1 < T) next_action = reinterpret_cast<const float4*>(actions)[(size_t)(t + 1) * a.n + i];
      if (a.ref_mode != QD_REF_STATIC) moving_reference(a, i, e.num_steps, e.ref);
      float r;
      uint8_t tr;
      env_step<LOAD, SPEC>(a, i, e, action, wtile + lane * a.D, &r, &tr);
      __builtin_nontemporal_store(r, reward + (size_t)t * a.n + i);
      __builtin_nontemporal_store(tr, trunc + (size_t)t * a.n + i);
    }
    __builtin_amdgcn_wave_barrier();
    if (wave_base < a.n) {
      const int rows = min(64, a.n - wave_base);
      flush_obs_any<SPEC>(wtile, obs + ((size_t)t * a.n + wave_base) * a.D, rows, a.D);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (live) store_env(a, i, e);
}

__global__ __launch_bounds__(64) void k_pid_reset(KArgs a, const uint8_t* __restrict__ mask) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n || (mask && !mask[i])) return;
  PidState<float> c;
  pid_reset(c);
  store_pid(a, i, c);
}

template <bool LOAD>
__global__ __launch_bounds__(64) void k_pid_action(KArgs a, float* __restrict__ actions) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  EnvRegs e;
  load_env<LOAD, false, true>(a, i, e);
  PidState<float> c;
  load_pid(a, i, c);
  reinterpret_cast<float4*>(actions)[i] = pid_env_action(c, e);
  store_pid(a, i, c);
}

// T closed-loop steps per launch: controller and env state both stay in registers between steps
template <bool LOAD, int SPEC>
__global__ __launch_bounds__(64) void k_rollout_pid(KArgs a, int T, float* __restrict__ obs, float* __restrict__ reward,
                                                    uint8_t* __restrict__ trunc, float* __restrict__ actions_out) {
  __shared__ float tile[OBS_LDS_FLOATS];
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave_base = i - lane;
  const bool live = i < a.n;
  EnvRegs e;
  PidState<float> c;
  if (live) {
    load_env<LOAD, false>(a, i, e);
    load_pid(a, i, c);
  }
  for (int t = 0; t < T; t++) {
    if (live) {
      if (a.ref_mode != QD_REF_STATIC) moving_reference(a, i, e.num_steps, e.ref);
      const float4 action = pid_env_action(c, e);
      if (actions_out) reinterpret_cast<float4*>(actions_out)[(size_t)t * a.n + i] = action;
      float r;
      uint8_t tr;
      env_step<LOAD, SPEC>(a, i, e, action, tile + lane * a.D, &r, &tr);
      if (a.auto_reset && tr) pid_reset(c);  // a new episode starts with fresh controller objects
      __builtin_nontemporal_store(r, reward + (size_t)t * a.n + i);
      __builtin_nontemporal_store(tr, trunc + (size_t)t * a.n + i);
    }
    __builtin_amdgcn_wave_barrier();
    if (wave_base < a.n) {
      const int rows = min(64, a.n - wave_base);
      flush_obs_any<SPEC>(tile, obs + ((size_t)t * a.n + wave_base) * a.D, rows, a.D);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (live) {
    store_env(a, i, e);
    store_pid(a, i, c);
  }
}

// _get_obs() of the current state for every env (also used after reset / regen)
template <bool LOAD, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_observe(KArgs a, float* __restrict__ obs) {
  __shared__ float tile[(BLOCK / 64) * OBS_LDS_FLOATS];
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* wtile = tile + wave * OBS_LDS_FLOATS;
  const int wave_base = i - lane;
  if (i < a.n) {
    EnvRegs e;
    load_env<LOAD>(a, i, e);
    if (e.flags & FLAG_ACC_STALE) refresh_sensor<LOAD>(a, e);
    float sv[33];
    M3<float> Rq;
    drone_state<float, LOAD>(e.s, e.acc, e.ref, e.par, sv, &Rq);
    write_obs_row<LOAD, SPEC_GENERIC>(a, e, sv, &Rq, wtile + lane * a.D);
  }
  __builtin_amdgcn_wave_barrier();
  if (wave_base < a.n) flush_obs(wtile, obs + (size_t)wave_base * a.D, min(64, a.n - wave_base), a.D);
}

// reset_model / reset_at: mask == nullptr && index < 0 -> all envs
template <bool LOAD>
__global__ __launch_bounds__(64) void k_reset(KArgs a, const uint8_t* __restrict__ mask, int index) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (index >= 0) i = (i == 0) ? index : a.n;
  if (i >= a.n) return;
  if (mask && !mask[i]) return;
  EnvRegs e;
  load_env<LOAD>(a, i, e);
  resample<LOAD>(a, i, e, true);
  store_env(a, i, e);
}

// generate_drone_params / explicit params -> raw planes, float32 copies, derived constants
__global__ __launch_bounds__(64) void k_params(KArgs a, ParamCfg pc, uint32_t regen, const double* __restrict__ explicit_raw,
                                               int fresh_data) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  double raw[6];
  if (explicit_raw) {
#pragma unroll
    for (int k = 0; k < 6; k++) raw[k] = explicit_raw[(size_t)i * 6 + k];
  } else {
    gen_params(pc, a.seed, (uint32_t)i, regen, raw);
  }
#pragma unroll
  for (int k = 0; k < 6; k++) a.raw[(size_t)k * a.npad + i] = raw[k];
  bool load;
  const Model<double> M = derive_model(raw, &load);
  float4* g = a.g;
  const int np = a.npad;
  const double* mp = reinterpret_cast<const double*>(&M);
#pragma unroll
  for (int k = 0; k < MODEL_FLOATS / 4; k++)
    g[(G_M0 + k) * np + i] = make_float4((float)mp[4 * k], (float)mp[4 * k + 1], (float)mp[4 * k + 2], (float)mp[4 * k + 3]);
  g[G_P0 * np + i] = make_float4((float)raw[0], (float)raw[1], (float)raw[2], (float)raw[3]);
  g[G_P1 * np + i] = make_float4((float)raw[4], (float)raw[5], (float)M.I2a, 0.f);
  // A reset-pool entry's second stage (the accelerometer's affine form, NXA* / NYA*) was evaluated with the model that has just
  // been replaced: back to "state only" -- the pre-sampled state itself does not depend on the model -- and a request, so that the
  // reset that follows (resample -> pool_fill) or the next launch's samplers redo the stage with the new constants.  Without
  // this the first row of the episode after a regeneration carried the previous parameter set's sensor reading.
  if (a.use_pool && a.obs_needs_acc) {
    bool downgraded = false;
#pragma unroll
    for (int slot = 0; slot < 2; slot++) {
      const int tp = (slot ? (int)G_NY4 : (int)G_NX4) * np + i;
      const float4 t = g[tp];
      if (__float_as_uint(t.z) == POOL_FULL) { g[tp] = make_float4(t.x, t.y, __uint_as_float(POOL_STATE), 0.f); downgraded = true; }
    }
    if (downgraded) pool_request(a, i);
  }
  if (fresh_data) {  // a new MjData: activations and sensordata start at zero
    g[G_ACT * np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
    g[G_ACC * np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// MjData() / mj_resetData: qpos0 from the spawn grid (env_gen.py:116-124), everything else zero.
// full != 0 additionally clears the per-env episode counters and per-env references (construction).
__global__ __launch_bounds__(64) void k_init_state(KArgs a, int full) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  const int sz = (int)ceil(sqrt((double)a.n));
  const double x = round5(((double)(i % sz) - (sz - 1) * 0.5) * 0.5);
  const double y = round5(((double)(i / sz) - (sz - 1) * 0.5) * 0.5);
  float4* g = a.g;
  const int np = a.npad;
  g[G_POS * np + i] = make_float4((float)x, (float)y, 0.15f, 0.f);
  g[G_QUAT * np + i] = make_float4(1.f, 0.f, 0.f, 0.f);
  g[G_VEL * np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
  g[G_ANG * np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
  g[G_ACT * np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
  g[G_ACC * np + i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (full) {
    g[G_AUX * np + i] = make_float4(0.f, __int_as_float(0), __uint_as_float(0u), 0.f);
    g[G_REF * np + i] = make_float4(a.ref[0], a.ref[1], a.ref[2], a.ref[3]);
    g[G_NX4 * np + i] = make_float4(0.f, 0.f, __uint_as_float(0u), 0.f);
    g[G_NY4 * np + i] = make_float4(0.f, 0.f, __uint_as_float(0u), 0.f);
    if ((i & 63) == 0) a.need[i >> 6] = 1u;   // empty pool: the first step launch's samplers fill it
    if (i == 0) { a.need[(a.npad >> 6) + 0] = 0u; a.need[(a.npad >> 6) + 1] = 0u; }
    PidState<float> c;
    pid_reset(c);
    store_pid(a, i, c);
  } else {
    float4 aux = g[G_AUX * np + i];
    aux.x = 0.f;
    aux.w = __uint_as_float(0u);
    g[G_AUX * np + i] = aux;
  }
}

__global__ __launch_bounds__(64) void k_get_params(KArgs a, double* __restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
#pragma unroll
  for (int k = 0; k < 6; k++) out[(size_t)i * 6 + k] = a.raw[(size_t)k * a.npad + i];
}

__global__ __launch_bounds__(64) void k_set_ref(KArgs a, const float* __restrict__ ref) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  a.g[G_REF * a.npad + i] = reinterpret_cast<const float4*>(ref)[i];
}

template <bool LOAD>
__global__ __launch_bounds__(64) void k_set_state(KArgs a, const float* __restrict__ qpos, const float* __restrict__ qvel,
                                                  const float* __restrict__ act) {
  constexpr int NQ = LOAD ? 9 : 7, NV = LOAD ? 8 : 6;
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  EnvRegs e;
  load_env<LOAD>(a, i, e);
  const float* qp = qpos + (size_t)i * NQ;
  const float* qv = qvel + (size_t)i * NV;
  e.s.px = qp[0]; e.s.py = qp[1]; e.s.pz = qp[2]; e.s.qw = qp[3]; e.s.qx = qp[4]; e.s.qy = qp[5]; e.s.qz = qp[6];
  e.s.vx = qv[0]; e.s.vy = qv[1]; e.s.vz = qv[2]; e.s.wx = qv[3]; e.s.wy = qv[4]; e.s.wz = qv[5];
  if (LOAD) { e.s.th1 = qp[7]; e.s.th2 = qp[8]; e.s.thd1 = qv[6]; e.s.thd2 = qv[7]; }
  if (act) { e.s.a0 = act[4 * i]; e.s.a1 = act[4 * i + 1]; e.s.a2 = act[4 * i + 2]; e.s.a3 = act[4 * i + 3]; }
  refresh_sensor<LOAD>(a, e);
  store_env(a, i, e);
}

template <bool LOAD>
__global__ __launch_bounds__(64) void k_get_state(KArgs a, float* __restrict__ qpos, float* __restrict__ qvel,
                                                  float* __restrict__ act, float* __restrict__ sens,
                                                  int32_t* __restrict__ num_steps) {
  constexpr int NQ = LOAD ? 9 : 7, NV = LOAD ? 8 : 6;
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  EnvRegs e;
  load_env<LOAD>(a, i, e);
  if (sens && (e.flags & FLAG_ACC_STALE)) refresh_sensor<LOAD>(a, e);
  if (qpos) {
    float* qp = qpos + (size_t)i * NQ;
    qp[0] = e.s.px; qp[1] = e.s.py; qp[2] = e.s.pz; qp[3] = e.s.qw; qp[4] = e.s.qx; qp[5] = e.s.qy; qp[6] = e.s.qz;
    if (LOAD) { qp[7] = e.s.th1; qp[8] = e.s.th2; }
  }
  if (qvel) {
    float* qv = qvel + (size_t)i * NV;
    qv[0] = e.s.vx; qv[1] = e.s.vy; qv[2] = e.s.vz; qv[3] = e.s.wx; qv[4] = e.s.wy; qv[5] = e.s.wz;
    if (LOAD) { qv[6] = e.s.thd1; qv[7] = e.s.thd2; }
  }
  if (act) { act[4 * i] = e.s.a0; act[4 * i + 1] = e.s.a1; act[4 * i + 2] = e.s.a2; act[4 * i + 3] = e.s.a3; }
  if (sens) { sens[3 * i] = e.acc.x; sens[3 * i + 1] = e.acc.y; sens[3 * i + 2] = e.acc.z; }
  if (num_steps) num_steps[i] = e.num_steps;
}

template <bool LOAD>
__global__ __launch_bounds__(64) void k_drone_states(KArgs a, float* __restrict__ out) {
  constexpr int NS = LOAD ? 33 : 29;
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  EnvRegs e;
  load_env<LOAD>(a, i, e);
  if (e.flags & FLAG_ACC_STALE) refresh_sensor<LOAD>(a, e);
  float sv[33];
  drone_state<float, LOAD>(e.s, e.acc, e.ref, e.par, sv);
#pragma unroll
  for (int k = 0; k < NS; k++) out[(size_t)i * NS + k] = sv[k];
}

// ---- stateless evaluation of the reference's pure functions ----------------------------
struct EvalArgs {
  float ref[4];
  float max_distance;
  int max_steps, kind, n, D;
};

template <int NS>
__global__ __launch_bounds__(64) void k_eval_obs(EvalArgs a, const float* __restrict__ states, float* __restrict__ obs) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  float sv[33];
#pragma unroll
  for (int k = 0; k < NS; k++) sv[k] = states[(size_t)i * NS + k];
  float* row = obs + (size_t)i * a.D;
#define QD_CALL(K) obs_to_lds<NS, K>(sv, a.ref, row, nullptr)
  QD_OBS_DISPATCH(a.kind, QD_CALL)
#undef QD_CALL
}

template <int NS>
__global__ __launch_bounds__(64) void k_eval_reward(EvalArgs a, const float* __restrict__ states,
                                                    const float* __restrict__ actions, const int32_t* __restrict__ ks,
                                                    float* __restrict__ out, uint8_t* __restrict__ tr) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  float sv[33];
#pragma unroll
  for (int k = 0; k < 33; k++) sv[k] = (k < NS) ? states[(size_t)i * NS + k] : 0.f;
  const int k = ks ? ks[i] : 0;
  if (out) {
    const float act4[4] = {actions[4 * i], actions[4 * i + 1], actions[4 * i + 2], actions[4 * i + 3]};
    out[i] = reward<float>(a.kind, sv, act4, k, a.ref, a.max_distance);
  }
  if (tr) tr[i] = truncated<float>(sv, a.ref, k, a.max_distance, a.max_steps) ? 1 : 0;
}

__global__ __launch_bounds__(64) void k_transform(int which, const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  if (which == QD_TF_QUAT2RPY || which == QD_TF_QUAT2DCM) {
    const float* q = in + 4 * i;
    const float qn = 1.f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float w = q[0] * qn, x = q[1] * qn, y = q[2] * qn, z = q[3] * qn;
    if (which == QD_TF_QUAT2RPY) {
      quat2rpy(w, x, y, z, out + 3 * i, out + 3 * i + 1, out + 3 * i + 2);
    } else {
      const M3<float> R = quat2mat(w, x, y, z);
      float* o = out + 9 * i;
      o[0] = R.m00; o[1] = R.m01; o[2] = R.m02; o[3] = R.m10; o[4] = R.m11; o[5] = R.m12; o[6] = R.m20; o[7] = R.m21; o[8] = R.m22;
    }
  } else if (which == QD_TF_RPY2QUAT || which == QD_TF_PENDRP2QUAT) {
    float* o = out + 4 * i;
    if (which == QD_TF_RPY2QUAT) {
      const float* r = in + 3 * i;
      float sr, cr, sp, cp, sy, cy;
      qsincos(0.5f * r[0], &sr, &cr); qsincos(0.5f * r[1], &sp, &cp); qsincos(0.5f * r[2], &sy, &cy);
      o[0] = cr * cp * cy + sr * sp * sy; o[1] = sr * cp * cy - cr * sp * sy;
      o[2] = cr * sp * cy + sr * cp * sy; o[3] = cr * cp * sy - sr * sp * cy;
    } else {
      const float* r = in + 2 * i;  // R = Rx(a) Ry(b) (transformation.py:27-29)
      float sa, ca, sb, cb;
      qsincos(0.5f * r[0], &sa, &ca); qsincos(0.5f * r[1], &sb, &cb);
      o[0] = ca * cb; o[1] = sa * cb; o[2] = ca * sb; o[3] = sa * sb;
    }
  } else if (which == QD_TF_DCM2QUAT) {
    const float* R = in + 9 * i;  // scipy from_matrix: largest of (R00, R11, R22, trace) picks the branch
    const float tr = R[0] + R[4] + R[8];
    float w, x, y, z;
    if (tr >= R[0] && tr >= R[4] && tr >= R[8]) {
      x = R[7] - R[5]; y = R[2] - R[6]; z = R[3] - R[1]; w = 1.f + tr;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {
      x = 1.f - tr + 2.f * R[0]; y = R[3] + R[1]; z = R[6] + R[2]; w = R[7] - R[5];
    } else if (R[4] >= R[8]) {
      y = 1.f - tr + 2.f * R[4]; z = R[7] + R[5]; x = R[1] + R[3]; w = R[2] - R[6];
    } else {
      z = 1.f - tr + 2.f * R[8]; x = R[2] + R[6]; y = R[5] + R[7]; w = R[3] - R[1];
    }
    const float nn = 1.f / sqrtf(w * w + x * x + y * y + z * z);
    float* o = out + 4 * i;
    o[0] = w * nn; o[1] = x * nn; o[2] = y * nn; o[3] = z * nn;
  }
}

}  // namespace qd

// =========================================================================== C ABI
using namespace qd;

struct qd_env {
  qd_config cfg;
  KArgs ka;
  ParamCfg pc;
  uint32_t regen;
  int D, ns;
  int spec;
  bool load;
  // qd_step_fragment: the T per-step launches of one fragment, captured once in a HIP graph and replayed; a few graphs are
  // kept (double-buffered fragments alternate between two sets of buffers)
  struct Frag {
    hipGraphExec_t exec = nullptr;
    const float* actions = nullptr;
    float* obs = nullptr;
    float* reward = nullptr;
    uint8_t* trunc = nullptr;
    int T = 0;
    unsigned long long used = 0;
  };
  static constexpr int FRAGS = 8;
  Frag frag[FRAGS];
  unsigned long long frag_clock = 0;
  hipStream_t frag_stream = nullptr;
  // qd_set_reference_schedule: waypoint k is the reference of the k-th step of the next policy rollout
  std::vector<double> ref_schedule;
  int opt[QD_OPT_COUNT] = {1, 1};   // qd_set_option
};

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
#define QD_HIP(call)                                                                           \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) return fail(QD_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));     \
  } while (0)
#define QD_LAUNCH_CHECK() QD_HIP(hipGetLastError())
// hipGetLastError() is sticky per thread: drop whatever an unrelated earlier HIP call of the host
// application left behind before launching, so that the check reports THIS launch only
#define QD_LAUNCH(...)                 \
  do {                                 \
    (void)hipGetLastError();           \
    hipLaunchKernelGGL(__VA_ARGS__);   \
  } while (0)
#define QD_NEED(env) \
  if (!(env)) return fail(QD_ERR_INVALID, "null env handle")

static inline int blocks64(int n) { return (n + 63) / 64; }
// batch size from which the 256-thread (two waves per SIMD, re-folded coefficients) step kernel is used.
// Measured crossover on MI355X: 65536 envs 8.6 us (64-thread) vs 9.3 us (256-thread), 131072 envs 14.4 vs 12.6 us.
// QD_BLOCK_THRESHOLD overrides it for experiments.
static int qd_block_threshold() {
  static const int v = [] { const char* e = getenv("QD_BLOCK_THRESHOLD"); return e ? atoi(e) : 98304; }();
  return v;
}
// fragments shorter than this are issued as direct launches instead of a captured graph (QD_GRAPH_MIN_STEPS overrides)
static int qd_graph_min_steps() {
  static const int v = [] { const char* e = getenv("QD_GRAPH_MIN_STEPS"); return e ? atoi(e) : 128; }();
  return v;
}
// largest batch stepped by the three-wave cooperative kernel (3 waves per 64 envs must find idle SIMDs; measured against the
// single-wave kernel: 5.13 / 5.40 us at 16384 envs, 5.95 / 6.25 at 24576, 10.1 / 7.1 at 32768); QD_COOP_MAX_ENVS overrides it
// (0 switches the kernel off)
static int qd_coop_max_envs() {
  static const int v = [] { const char* e = getenv("QD_COOP_MAX_ENVS"); return e ? atoi(e) : 24576; }();
  return v;
}
// batches below this size run reset-pool sampler workgroups next to the stepping ones (as many again for the 64-thread kernels:
// from 32768 envs the two together no longer find a SIMD each -- 49152 envs: 10.8 us per step with samplers, 7.4 with the
// resets sampled in the stepping waves; 16384: 5.4 against 6.1); QD_POOL_MAX_ENVS overrides it
static int qd_pool_max_envs() {
  static const int v = [] { const char* e = getenv("QD_POOL_MAX_ENVS"); return e ? atoi(e) : 32768; }();
  return v;
}
static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
// qd_step_fragment / qd_rollout of the training configuration at small batches run as ONE persistent launch (k_rollout_coop,
// qd_rollout_coop.hip); QD_PERSISTENT=0 keeps the per-step launches (graph replay), for comparisons
static int qd_persistent() {
  static const int v = [] { const char* e = getenv("QD_PERSISTENT"); return e ? atoi(e) : 1; }();
  return v;
}

extern "C" {

const char* qd_last_error(void) { return g_err; }
int qd_version(void) { return QD_VERSION; }
#ifdef QD_STAMPS
int qd_debug_read_stamps(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(qd_stamps), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? 0 : -4;
}
int qd_debug_read_pstamps(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(qd::qd_pstamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -4;
}
int qd_debug_read_cstamps(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(qd_cstamps), sizeof(unsigned long long) * 64 * 3 * 16) == hipSuccess ? 0 : -4;
}
int qd_debug_read_rstamps(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(qd_rstamps), sizeof(unsigned long long) * 64 * 2) == hipSuccess ? 0 : -4;
}
#endif

int qd_state_dim(int model) { return model == QD_MODEL_LOAD ? 33 : 29; }

int qd_obs_dim(int obs_kind, int model) {
  if (obs_kind < 0 || obs_kind >= QD_OBS_COUNT) return -1;
  return obs_dim(obs_kind, qd_state_dim(model));
}

static inline int npad_of(int n) { return (n + PAD - 1) / PAD * PAD; }

size_t qd_arena_bytes(int num_envs) {
  if (num_envs <= 0) return 0;
  const size_t np = (size_t)npad_of(num_envs);
  return np * ((size_t)NUM_GROUPS * sizeof(float4) + (size_t)RAW_PLANES * sizeof(double)) + (np / 64 + POOL_STAT_WORDS) * sizeof(uint32_t);
}

static int needs_load_reward(int k) {
  return (k >= QD_REW_PEND_EN && k <= QD_REW_PEND_EN4) || (k >= QD_REW_PEND_DIST && k <= QD_REW_REWARD_3);
}

int qd_create(const qd_config* c, void* arena, size_t arena_bytes, qd_env** out) {
  if (!c || !out) return fail(QD_ERR_INVALID, "null argument");
  if (c->num_envs <= 0) return fail(QD_ERR_INVALID, "num_envs must be positive");
  if (c->model != QD_MODEL_LOAD && c->model != QD_MODEL_NOLOAD) return fail(QD_ERR_INVALID, "unknown model %d", c->model);
  if (c->obs_kind < 0 || c->obs_kind >= QD_OBS_COUNT) return fail(QD_ERR_INVALID, "unknown observation variant %d", c->obs_kind);
  if (c->obs_kind == QD_OBS_PRY_ACC_PARAMS_NOPEND)
    return fail(QD_ERR_UNSUPPORTED, "LocalFramePRYaccParamsNoPendEnv raises NameError('acc') in the reference "
                                    "(observation_wrappers.py:448); it has no defined output");
  if (c->reward_kind < 0 || c->reward_kind >= QD_REW_COUNT) return fail(QD_ERR_INVALID, "unknown reward %d", c->reward_kind);
  if (c->model == QD_MODEL_NOLOAD && needs_load_reward(c->reward_kind))
    return fail(QD_ERR_UNSUPPORTED, "reward %d indexes params[4]/[5] of the state vector, which raises IndexError "
                                    "on the 29-element no-load state in the reference", c->reward_kind);
  if (c->obs_kind == QD_OBS_SIMPLE && c->model != QD_MODEL_NOLOAD)
    return fail(QD_ERR_UNSUPPORTED, "SimpleDrone observation requires the no-load model");
  if ((c->obs_kind == QD_OBS_SIMPLE) != (c->term_kind == QD_TERM_SIMPLE))
    return fail(QD_ERR_INVALID, "the SimpleDrone observation and the SimpleDrone termination rule go together");
  if (c->random_start < 0 || c->random_start > QD_START_SIMPLE) return fail(QD_ERR_INVALID, "unknown random_start %d", c->random_start);
  if (c->random_start == QD_START_SIMPLE && c->model != QD_MODEL_NOLOAD)
    return fail(QD_ERR_UNSUPPORTED, "QD_START_SIMPLE is SimpleDrone.reset_model: it exists for the no-load model only");
  if (c->frame_skip < 1) return fail(QD_ERR_INVALID, "frame_skip must be >= 1");
  if (!(c->timestep > 0)) return fail(QD_ERR_INVALID, "timestep must be positive");
  if (!arena) return fail(QD_ERR_ARENA, "null arena");
  if (reinterpret_cast<uintptr_t>(arena) % 256 != 0) return fail(QD_ERR_ARENA, "arena must be 256-byte aligned");
  if (arena_bytes < qd_arena_bytes(c->num_envs))
    return fail(QD_ERR_ARENA, "arena has %zu bytes, need %zu", arena_bytes, qd_arena_bytes(c->num_envs));
  qd_env* e = new (std::nothrow) qd_env();  // value-initialised: the plain members are zeroed, the vector and the graph slots default-constructed
  if (!e) return fail(QD_ERR_INVALID, "out of host memory");
  e->cfg = *c;
  e->load = c->model == QD_MODEL_LOAD;
  e->ns = qd_state_dim(c->model);
  e->D = qd_obs_dim(c->obs_kind, c->model);
  KArgs& k = e->ka;
  k.n = c->num_envs;
  k.npad = npad_of(c->num_envs);
  k.g = reinterpret_cast<float4*>(arena);
  k.raw = reinterpret_cast<double*>(k.g + (size_t)NUM_GROUPS * k.npad);
  k.need = reinterpret_cast<uint32_t*>(k.raw + (size_t)RAW_PLANES * k.npad);
  for (int i = 0; i < 4; i++) k.ref[i] = (float)c->reference[i];
  k.per_env_ref = c->per_env_reference;
  k.h = (float)c->timestep;
  k.frame_skip = c->frame_skip; k.ctrl_map = c->ctrl_map; k.obs_kind = c->obs_kind; k.reward_kind = c->reward_kind;
  k.term_kind = c->term_kind; k.max_distance = (float)c->max_distance; k.max_steps = c->max_steps;
  k.auto_reset = c->auto_reset; k.D = e->D; k.seed = c->seed;
  // sampler workgroups pay off while the launch is a latency chain (few waves, idle CUs); with >= 32768 envs the
  // chip is full and a second set of workgroups only adds traffic, so truncated lanes sample inline there
  if (c->ref_mode < QD_REF_STATIC || c->ref_mode > QD_REF_RAMP) { delete e; return fail(QD_ERR_INVALID, "unknown ref_mode %d", c->ref_mode); }
  k.ref_k0 = 0; k.ref_kmax = 0; k.ref_dt = k.ref_t0 = k.ref_inv_span = 0.f;
  for (int i = 0; i < 4; i++) k.ref_end[i] = (float)c->ref_end[i];
  if (c->ref_mode == QD_REF_STEP || c->ref_mode == QD_REF_RAMP) {
    const double dt = c->timestep * c->frame_skip, t0 = c->ref_t0, dur = c->ref_duration;
    if (!(dur > 0.0) || !(t0 >= 0.0) || (c->ref_mode == QD_REF_RAMP && !(dur > t0)) || dur / dt > 1e9) {
      delete e;
      return fail(QD_ERR_INVALID, "step / ramp reference needs 0 <= ref_t0 (< ref_duration for a ramp), ref_duration > 0");
    }
    // the comparisons numpy makes on t = arange(0, duration, dt): t_k = k * dt in float64
    long long k0 = (long long)ceil(t0 / dt);
    while (k0 > 0 && (double)(k0 - 1) * dt >= t0) k0--;
    while ((double)k0 * dt < t0) k0++;
    const long long len = (long long)ceil(dur / dt);
    k.ref_k0 = (int)(k0 < 2000000000LL ? k0 : 2000000000LL);
    k.ref_kmax = (int)(len > 0 ? len - 1 : 0);
    k.ref_dt = (float)dt; k.ref_t0 = (float)t0;
    k.ref_inv_span = c->ref_mode == QD_REF_RAMP ? (float)(1.0 / (dur - t0)) : 0.f;
  }
  k.ref_mode = c->ref_mode;
  k.ref_radius = (float)c->ref_radius;
  k.ref_omega_dt = (float)(6.283185307179586 * c->ref_frequency * c->timestep * c->frame_skip);
  k.ref_phase_step = (float)(6.283185307179586 / c->num_envs);
  k.use_pool = (c->auto_reset && c->random_start != QD_START_FIXED && c->num_envs < qd_pool_max_envs()) ? 1 : 0;
  k.main_blocks = 0;
  {
    const int ok = c->obs_kind;
    const bool reads_acc_load = ok == QD_OBS_RAW || ok == QD_OBS_FULLSTATE || ok == QD_OBS_FULLSTATE_ZVEC || ok == QD_OBS_PRY_ACC ||
                                ok == QD_OBS_PRY_ACC_PARAMS || ok == QD_OBS_PRY_ACC_NOPEND;
    // without the load the wrappers' "pendulum" slices alias the accelerometer entries (state[12:15])
    k.obs_needs_acc = e->load ? (reads_acc_load ? 1 : 0) : (ok != QD_OBS_SIMPLE ? 1 : 0);
  }
  SampleCfg& sc = k.sc;
  for (int i = 0; i < 4; i++) sc.start_pos[i] = (float)c->start_pos[i];
  sc.max_pos_offset = (float)c->max_pos_offset;
  for (int i = 0; i < 2; i++) { sc.angle_var[i] = (float)c->angle_var[i]; sc.pend_rp_var[i] = (float)c->pend_rp_var[i]; sc.pend_vel_var[i] = (float)c->pend_vel_var[i]; }
  for (int i = 0; i < 3; i++) { sc.vel_var[i] = (float)c->vel_var[i]; sc.ang_vel_var[i] = (float)c->ang_vel_var[i]; }
  sc.random_start = c->random_start;
  for (int i = 0; i < 6; i++) { e->pc.center[i] = c->param_center[i]; e->pc.width[i] = c->param_width[i]; }
  e->pc.difficulty = c->param_difficulty;
  e->pc.random_params = c->random_params;
  e->pc.load = e->load ? 1 : 0;
  e->regen = 0;
  e->spec = SPEC_GENERIC;
  if (c->floor_contact) {
    e->spec = SPEC_FLOOR;
  } else if (e->load && c->obs_kind == QD_OBS_RPY_PARAMS && c->reward_kind == QD_REW_DISTANCE_ENERGY && c->ctrl_map == QD_CTRL_AFFINE &&
      c->term_kind == QD_TERM_DEFAULT && c->frame_skip == 1)
    e->spec = SPEC_RMA;
  else if (e->load && c->obs_kind == QD_OBS_FULLSTATE && c->reward_kind == QD_REW_PEND_EN4 && c->ctrl_map == QD_CTRL_AFFINE &&
           c->term_kind == QD_TERM_DEFAULT && c->frame_skip == 1)
    e->spec = SPEC_LSTM;
  else if (!e->load && c->obs_kind == QD_OBS_SIMPLE && c->term_kind == QD_TERM_SIMPLE && c->ctrl_map == QD_CTRL_DIRECT &&
           c->frame_skip == 2)
    e->spec = SPEC_SIMPLE;
  else if (c->frame_skip == 1)
    e->spec = SPEC_GENERIC_FS1;
  *out = e;
  return QD_OK;
}

int qd_destroy(qd_env* env) {
  if (env) {
    for (auto& f : env->frag)
      if (f.exec) (void)hipGraphExecDestroy(f.exec);
    if (env->frag_stream) (void)hipStreamDestroy(env->frag_stream);
  }
  delete env;
  return QD_OK;
}

int qd_init(qd_env* env, void* stream) {
  QD_NEED(env);
  const KArgs& k = env->ka;
  env->regen = 0;
  QD_LAUNCH(k_init_state, dim3(blocks64(k.n)), dim3(64), 0, S(stream), k, 1);
  QD_LAUNCH_CHECK();
  QD_LAUNCH(k_params, dim3(blocks64(k.n)), dim3(64), 0, S(stream), k, env->pc, env->regen, (const double*)nullptr, 1);
  QD_LAUNCH_CHECK();
  if (env->spec == SPEC_FLOOR) QD_HIP(launch_floor_consts(k, env->load, S(stream)));
  return QD_OK;
}

int qd_reset_data(qd_env* env, void* stream) {
  QD_NEED(env);
  QD_LAUNCH(k_init_state, dim3(blocks64(env->ka.n)), dim3(64), 0, S(stream), env->ka, 0);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

int qd_set_reference(qd_env* env, const double ref_host[4]) {
  QD_NEED(env);
  if (!ref_host) return fail(QD_ERR_INVALID, "null reference");
  for (int i = 0; i < 4; i++) { env->cfg.reference[i] = ref_host[i]; env->ka.ref[i] = (float)ref_host[i]; }
  for (auto& f : env->frag) f.T = 0;  // captured fragment graphs hold the old reference in their kernel arguments
  return QD_OK;
}

int qd_set_reference_schedule(qd_env* env, const double* traj_host, int T) {
  QD_NEED(env);
  if (T < 0 || (T > 0 && !traj_host)) return fail(QD_ERR_INVALID, "bad reference schedule");
  env->ref_schedule.assign(traj_host, traj_host + (size_t)4 * T);
  return QD_OK;
}

int qd_set_reference_per_env(qd_env* env, const float* ref, void* stream) {
  QD_NEED(env);
  if (!env->ka.per_env_ref) return fail(QD_ERR_INVALID, "env was created without per_env_reference");
  if (!ref) return fail(QD_ERR_INVALID, "null reference array");
  QD_LAUNCH(k_set_ref, dim3(blocks64(env->ka.n)), dim3(64), 0, S(stream), env->ka, ref);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

int qd_randomize_params(qd_env* env, void* stream) {
  QD_NEED(env);
  env->regen += 1;
  QD_LAUNCH(k_params, dim3(blocks64(env->ka.n)), dim3(64), 0, S(stream), env->ka, env->pc, env->regen,
                     (const double*)nullptr, 1);
  QD_LAUNCH_CHECK();
  if (env->spec == SPEC_FLOOR) QD_HIP(launch_floor_consts(env->ka, env->load, S(stream)));
  return QD_OK;
}

int qd_set_params(qd_env* env, const double* raw, void* stream) {
  QD_NEED(env);
  if (!raw) return fail(QD_ERR_INVALID, "null params");
  QD_LAUNCH(k_params, dim3(blocks64(env->ka.n)), dim3(64), 0, S(stream), env->ka, env->pc, env->regen, raw, 0);
  QD_LAUNCH_CHECK();
  if (env->spec == SPEC_FLOOR) QD_HIP(launch_floor_consts(env->ka, env->load, S(stream)));
  return QD_OK;
}

int qd_get_params(qd_env* env, double* raw, void* stream) {
  QD_NEED(env);
  if (!raw) return fail(QD_ERR_INVALID, "null output");
  QD_LAUNCH(k_get_params, dim3(blocks64(env->ka.n)), dim3(64), 0, S(stream), env->ka, raw);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

#define QD_BY_MODEL(env, KERNEL, grid, block, stream, ...)                                            \
  do {                                                                                                \
    if ((env)->load) QD_LAUNCH((KERNEL<true>), grid, block, 0, S(stream), __VA_ARGS__);      \
    else QD_LAUNCH((KERNEL<false>), grid, block, 0, S(stream), __VA_ARGS__);                 \
    QD_LAUNCH_CHECK();                                                                                \
  } while (0)

static int launch_observe(qd_env* env, float* obs, void* stream) {
  const KArgs& k = env->ka;
  if (env->load) QD_LAUNCH((k_observe<true, 64>), dim3(blocks64(k.n)), dim3(64), 0, S(stream), k, obs);
  else QD_LAUNCH((k_observe<false, 64>), dim3(blocks64(k.n)), dim3(64), 0, S(stream), k, obs);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

int qd_reset(qd_env* env, const uint8_t* mask, float* obs, void* stream) {
  QD_NEED(env);
  const KArgs& k = env->ka;
  QD_BY_MODEL(env, k_reset, dim3(blocks64(k.n)), dim3(64), stream, k, mask, -1);
  if (obs) return launch_observe(env, obs, stream);
  return QD_OK;
}

int qd_reset_at(qd_env* env, int index, void* stream) {
  QD_NEED(env);
  if (index < 0 || index >= env->ka.n) return fail(QD_ERR_INDEX, "index %d out of range [0, %d)", index, env->ka.n);
  QD_BY_MODEL(env, k_reset, dim3(1), dim3(64), stream, env->ka, (const uint8_t*)nullptr, index);
  return QD_OK;
}

int qd_set_state(qd_env* env, const float* qpos, const float* qvel, const float* act, void* stream) {
  QD_NEED(env);
  if (!qpos || !qvel) return fail(QD_ERR_INVALID, "qpos and qvel are required");
  QD_BY_MODEL(env, k_set_state, dim3(blocks64(env->ka.n)), dim3(64), stream, env->ka, qpos, qvel, act);
  return QD_OK;
}

int qd_get_state(qd_env* env, float* qpos, float* qvel, float* act, float* sensordata, int32_t* num_steps, void* stream) {
  QD_NEED(env);
  QD_BY_MODEL(env, k_get_state, dim3(blocks64(env->ka.n)), dim3(64), stream, env->ka, qpos, qvel, act, sensordata, num_steps);
  return QD_OK;
}

int qd_step(qd_env* env, const float* actions, int64_t n_action_values, float* obs, float* reward, uint8_t* truncated,
            void* stream) {
  QD_NEED(env);
  const KArgs& k = env->ka;
  if (n_action_values != (int64_t)4 * k.n) return fail(QD_ERR_SHAPE, "Action dimension mismatch");
  if (!actions || !obs || !reward || !truncated) return fail(QD_ERR_INVALID, "null array argument");
  // one wavefront per workgroup while the batch fits the chip once (1024 SIMDs x 64 lanes = 65536 envs: every wave
  // has a SIMD to itself); 256-thread workgroups capped at 256 registers beyond that (two waves per SIMD)
#define QD_STEP_LAUNCH_64(LOADV, SPECV)                                                                            \
  do {                                                                                                            \
    KArgs kk = k;                                                                                                 \
    kk.main_blocks = blocks64(k.n);                                                                               \
    QD_LAUNCH((k_step<LOADV, 64, SPECV>), dim3(kk.main_blocks * (k.use_pool ? 2 : 1)), dim3(64), 0, S(stream), kk, actions, obs, \
              reward, truncated);                                                                                 \
  } while (0)
#define QD_STEP_LAUNCH_256(LOADV, SPECV)                                                                           \
  do {                                                                                                            \
    KArgs kk = k;                                                                                                 \
    kk.main_blocks = (k.n + 255) / 256;                                                                           \
    QD_LAUNCH((k_step_wide<LOADV, 256, SPECV>), dim3(kk.main_blocks), dim3(256), 0, S(stream), kk.g, actions, kk.npad, kk.n,     \
              kk.main_blocks, kk, obs, reward, truncated);                                                        \
  } while (0)
#define QD_STEP_BLOCK(L)                                                     \
  do {                                                                       \
    if (env->load) {                                                         \
      if (env->spec == SPEC_RMA) L(true, SPEC_RMA);                          \
      else if (env->spec == SPEC_LSTM) L(true, SPEC_LSTM);                   \
      else if (env->spec == SPEC_GENERIC_FS1) L(true, SPEC_GENERIC_FS1);     \
      else L(true, SPEC_GENERIC);                                            \
    } else {                                                                 \
      if (env->spec == SPEC_SIMPLE) L(false, SPEC_SIMPLE);                   \
      else if (env->spec == SPEC_GENERIC_FS1) L(false, SPEC_GENERIC_FS1);    \
      else L(false, SPEC_GENERIC);                                           \
    }                                                                        \
  } while (0)
  if (env->spec == SPEC_FLOOR) {
    // the floor plane: the contact solve is a lane-group computation of the wavefront (k_step_floor, qd_step_floor.hip)
    QD_HIP(launch_step_floor(k, env->load, actions, obs, reward, truncated, S(stream)));
    return QD_OK;
  }
  if (env->load && env->spec == SPEC_RMA && k.n <= qd_coop_max_envs()) {
    // small batches of train_PPO.py / train_RMA.py's configuration: three wavefronts per 64 envs (k_step_coop)
    KArgs kk = k;
    kk.main_blocks = blocks64(k.n);
    const dim3 grid(kk.main_blocks + (k.use_pool ? (k.n + COOP_THREADS - 1) / COOP_THREADS : 0));
    QD_LAUNCH((k_step_coop<SPEC_RMA>), grid, dim3(COOP_THREADS), 0, S(stream), kk.g, actions, kk.npad, kk.n, kk.main_blocks, kk, obs, reward,
              truncated);
  } else if (k.n >= qd_block_threshold()) QD_STEP_BLOCK(QD_STEP_LAUNCH_256);
  else QD_STEP_BLOCK(QD_STEP_LAUNCH_64);
#undef QD_STEP_BLOCK
#undef QD_STEP_LAUNCH_64
#undef QD_STEP_LAUNCH_256
  QD_LAUNCH_CHECK();
  return QD_OK;
}

// does a fragment of this env run as one persistent launch (k_rollout_lat / k_rollout_coop)?  The load model with one substep per step,
// at every batch size: measured against the per-step launches on one box (tests/diag_persistent_big.py, profiles/r03_persistent_vs_per_step.txt)
// 4096 envs 1.50 / 3.99 us per step, 16384 1.60 / 5.12, 65536 5.19 / 7.71, 262144 20.5 / 23.5, 2^20 73.2 / 70.9, 2^22 249 / 289.
// QD_PERSISTENT_MAX_ENVS lowers the limit for experiments.
static int qd_persistent_max_envs() {
  static const int v = [] { const char* e = getenv("QD_PERSISTENT_MAX_ENVS"); return e ? atoi(e) : 0x7fffffff; }();
  return v;
}
// (the sensor-carrying / run-time-dispatched instantiations: measured with config 5 against its per-step kernels, steady state,
// 1.87 / 5.43 us per step at 8192 envs, 3.39 / 9.72 at 32768, 23.8 / 31.0 at 262144, 92.6 / 100.1 at 2^20; QD_PERSISTENT_MAX_ENVS_OTHER
// limits them separately for experiments)
static int qd_persistent_max_envs_other() {
  static const int v = [] { const char* e = getenv("QD_PERSISTENT_MAX_ENVS_OTHER"); return e ? atoi(e) : 0x7fffffff; }();
  return v;
}
static bool qd_fragment_is_persistent(const qd_env* env) {
  if (!(qd_persistent() && env->opt[QD_OPT_PERSISTENT_FRAGMENTS] && env->load)) return false;
  if (env->spec == SPEC_RMA) return env->ka.n <= qd_persistent_max_envs();
  if (env->spec == SPEC_LSTM || env->spec == SPEC_GENERIC_FS1) return env->ka.n <= qd_persistent_max_envs() && env->ka.n <= qd_persistent_max_envs_other();
  return false;
}
// ... and of those, the batches that leave every workgroup a CU of its own (<= 256 workgroups) in the latency arrangement of the
// same step (k_rollout_lat, qd_rollout_lat.hip): train_PPO.py / train_RMA.py's configuration and every run-time-dispatched one whose
// observation row does not carry the accelerometer
static bool qd_fragment_is_latency(const qd_env* env) {
  if (!qd_fragment_is_persistent(env) || !env->opt[QD_OPT_LATENCY_KERNEL]) return false;
  if (env->ka.n > 256 * 64) return false;
  return env->spec == SPEC_RMA || env->spec == SPEC_LSTM || env->spec == SPEC_GENERIC_FS1;
}
// The single-body model (SimpleDrone, and BaseDroneEnv without the load): its step is ~500 float32 instructions, too short to be
// worth a split over waves, so a fragment runs in k_rollout -- one wavefront per 64 envs, state in registers, rows through the
// wave's LDS tile.  Measured against the per-step launches (tests/diag_rollout_simple.py, BASELINE config 2): 1.57 / 3.51 us per
// step at 4096 envs, 2.03 / 5.13 at 65536, 24.8 / 54.7 at 2^20 (4.2e10 env-steps/s).
// SimpleDrone while every workgroup finds a CU of its own (<= 16384 envs): physics and epilogue in two wavefronts (k_rollout_pair);
// above, the chip is full either way and the single wavefront of k_rollout issues fewer instructions in all.  QD_PAIR_MAX_ENVS.
static bool qd_fragment_is_pair(const qd_env* env) {
  static const int v = [] { const char* e = getenv("QD_PAIR_MAX_ENVS"); return e ? atoi(e) : 16384; }();
  return qd_persistent() && env->opt[QD_OPT_PERSISTENT_FRAGMENTS] && !env->load && env->spec == SPEC_SIMPLE && env->ka.n <= v;
}
static bool qd_fragment_is_rollout(const qd_env* env) {
  return qd_persistent() && env->opt[QD_OPT_PERSISTENT_FRAGMENTS] && !env->load && env->spec != SPEC_FLOOR;
}

int qd_set_option(qd_env* env, int option, int value) {
  QD_NEED(env);
  if (option < 0 || option >= QD_OPT_COUNT) return fail(QD_ERR_INVALID, "unknown option %d", option);
  env->opt[option] = value;
  return QD_OK;
}

// the variant selector's answer, in words (the selection itself: qd_step / qd_step_fragment)
const char* qd_step_kernel_name(const qd_env* env) {
  if (!env) return "";
  static const char* const spec_name[] = {"0", "1", "2", "3", "4", "5"};
  static thread_local char buf[64];
  const int n = env->ka.n;
  if (env->spec == SPEC_FLOOR) return env->load ? "qd::k_step_floor<true>" : "qd::k_step_floor<false>";
  if (env->load && env->spec == SPEC_RMA && n <= qd_coop_max_envs()) return "qd::k_step_coop<1>";
  const bool wide = n >= qd_block_threshold();
  snprintf(buf, sizeof buf, "qd::%s<%s,%d,%s>", wide ? "k_step_wide" : "k_step", env->load ? "true" : "false", wide ? 256 : 64,
           spec_name[env->spec >= 0 && env->spec <= 5 ? env->spec : 0]);
  return buf;
}
const char* qd_fragment_kernel_name(const qd_env* env) {
  if (!env) return "";
  if (qd_fragment_is_pair(env)) return "qd::k_rollout_pair";
  if (qd_fragment_is_rollout(env)) {
    static thread_local char buf[64];
    snprintf(buf, sizeof buf, "qd::k_rollout<false,64,%d>", env->spec);
    return buf;
  }
  if (!qd_fragment_is_persistent(env)) return qd_step_kernel_name(env);
  if (qd_fragment_is_latency(env)) return env->spec == SPEC_RMA ? "qd::k_rollout_lat<1>" : env->spec == SPEC_LSTM ? "qd::k_rollout_lat<2>" : "qd::k_rollout_lat<4>";
  const bool two = env->ka.n > 256 * 64;
  return env->spec == SPEC_RMA ? "qd::k_rollout_coop<1,2>"
       : env->spec == SPEC_LSTM ? (two ? "qd::k_rollout_coop<2,2>" : "qd::k_rollout_coop<2,1>")
                                : (two ? "qd::k_rollout_coop<4,2>" : "qd::k_rollout_coop<4,1>");
}

int qd_step_fragment(qd_env* env, const float* actions, int T, float* obs, float* reward, uint8_t* truncated, void* stream) {
  QD_NEED(env);
  if (T < 0) return fail(QD_ERR_INVALID, "negative step count");
  if (T == 0) return QD_OK;
  if (!actions || !obs || !reward || !truncated) return fail(QD_ERR_INVALID, "null array argument");
  const int n = env->ka.n;
  const size_t D = (size_t)env->D;
  if (qd_fragment_is_latency(env)) {
    QD_HIP(launch_rollout_lat(env->ka, env->spec, T, actions, obs, reward, truncated, S(stream)));
    return QD_OK;
  }
  if (qd_fragment_is_persistent(env)) {
    QD_HIP(launch_rollout_coop(env->ka, env->spec, T, actions, obs, reward, truncated, S(stream)));
    return QD_OK;
  }
  if (qd_fragment_is_rollout(env)) return qd_rollout(env, actions, T, obs, reward, truncated, stream);
  qd_env::Frag* fr = nullptr;
  for (auto& f : env->frag)
    if (f.T == T && f.actions == actions && f.obs == obs && f.reward == reward && f.trunc == truncated) fr = &f;
  // A graph costs ~20 us per node to capture and instantiate (370 us for 20 steps, measured in round 1 inside a timed region)
  // and 10-16 us per replay; a direct launch costs the host 3.5-5 us, about the kernel's own period.  So: long runs are
  // captured at first sight; a short run (< 128 steps) is issued launch by launch the first time these buffers and this
  // length are seen, captured the second time (a caller that repeats it will repeat it again) and replayed from then on.
  const bool known = fr != nullptr, never = qd_graph_min_steps() >= (1 << 30);   // QD_GRAPH_MIN_STEPS >= 2^30: no graphs at all (counter passes)
  if (!(fr && fr->exec) && T < qd_graph_min_steps() && (!known || never)) {
    if (never) {
      for (int t = 0; t < T; t++) {
        const int rc = qd_step(env, actions + (size_t)t * n * 4, (int64_t)n * 4, obs + (size_t)t * n * D, reward + (size_t)t * n,
                               truncated + (size_t)t * n, stream);
        if (rc != QD_OK) return rc;
      }
      return QD_OK;
    }
    fr = &env->frag[0];
    for (auto& f : env->frag)
      if (f.T == 0) { fr = &f; break; } else if (f.used < fr->used) fr = &f;   // a free slot, else the least recently used
    if (fr->exec) { (void)hipGraphExecDestroy(fr->exec); fr->exec = nullptr; }
    fr->actions = actions; fr->obs = obs; fr->reward = reward; fr->trunc = truncated; fr->T = T;   // remembered, not captured
    fr->used = ++env->frag_clock;
    for (int t = 0; t < T; t++) {
      const int rc = qd_step(env, actions + (size_t)t * n * 4, (int64_t)n * 4, obs + (size_t)t * n * D, reward + (size_t)t * n,
                             truncated + (size_t)t * n, stream);
      if (rc != QD_OK) return rc;
    }
    return QD_OK;
  }
  if (!(fr && fr->exec)) {
    // capture on a private stream (the caller's may be the legacy default stream, which cannot be captured) ...
    if (!env->frag_stream) QD_HIP(hipStreamCreateWithFlags(&env->frag_stream, hipStreamNonBlocking));
    if (!fr) {
      fr = &env->frag[0];
      for (auto& f : env->frag)
        if (f.T == 0) { fr = &f; break; } else if (f.used < fr->used) fr = &f;   // a free slot, else the least recently used
    }
    if (fr->exec) { (void)hipGraphExecDestroy(fr->exec); fr->exec = nullptr; }
    fr->T = 0;
    QD_HIP(hipStreamBeginCapture(env->frag_stream, hipStreamCaptureModeThreadLocal));
    int rc = QD_OK;
    for (int t = 0; t < T && rc == QD_OK; t++)
      rc = qd_step(env, actions + (size_t)t * n * 4, (int64_t)n * 4, obs + (size_t)t * n * D, reward + (size_t)t * n,
                   truncated + (size_t)t * n, env->frag_stream);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(env->frag_stream, &graph);
    if (rc != QD_OK || e != hipSuccess || !graph) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc != QD_OK ? rc : fail(QD_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    }
    const hipError_t ei = hipGraphInstantiate(&fr->exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { fr->exec = nullptr; return fail(QD_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ei)); }
    fr->actions = actions; fr->obs = obs; fr->reward = reward; fr->trunc = truncated; fr->T = T;
  }
  fr->used = ++env->frag_clock;
  // ... and replay in the caller's stream, in order with everything else there.  (Running the graph on the private stream
  // between cross-stream event waits costs 0.9 us PER KERNEL on this runtime: 5.9 instead of 5.0 us per step.)
  QD_HIP(hipGraphLaunch(fr->exec, S(stream)));
  return QD_OK;
}

int qd_pool_counters(qd_env* env, uint32_t* counters, void* stream) {
  QD_NEED(env);
  if (!counters) return fail(QD_ERR_INVALID, "null output");
  QD_HIP(hipMemcpyAsync(counters, env->ka.need + (env->ka.npad >> 6), 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, S(stream)));
  return QD_OK;
}

int qd_health_counters(qd_env* env, uint32_t* counters, void* stream) {
  QD_NEED(env);
  if (!counters) return fail(QD_ERR_INVALID, "null output");
  QD_HIP(hipMemcpyAsync(counters, env->ka.need + (env->ka.npad >> 6) + 8, QD_HEALTH_COUNTERS * sizeof(uint32_t), hipMemcpyDeviceToDevice, S(stream)));
  return QD_OK;
}

int qd_rollout(qd_env* env, const float* actions, int T, float* obs, float* reward, uint8_t* truncated, void* stream) {
  QD_NEED(env);
  const KArgs& k = env->ka;
  if (T < 0) return fail(QD_ERR_INVALID, "negative step count");
  if (T == 0) return QD_OK;
  if (!actions || !obs || !reward || !truncated) return fail(QD_ERR_INVALID, "null array argument");
  if (env->spec == SPEC_FLOOR) {
    // floor contact: the contact solve wants the whole register file for itself; with the env state kept in registers across
    // steps on top of it the multi-step kernel spilled 135 registers, so these envs are stepped launch by launch (same results)
    const size_t D = (size_t)env->D;
    for (int t = 0; t < T; t++) {
      const int rc = qd_step(env, actions + (size_t)t * k.n * 4, (int64_t)k.n * 4, obs + (size_t)t * k.n * D, reward + (size_t)t * k.n,
                             truncated + (size_t)t * k.n, stream);
      if (rc != QD_OK) return rc;
    }
    return QD_OK;
  }
  if (qd_fragment_is_latency(env)) {
    QD_HIP(launch_rollout_lat(k, env->spec, T, actions, obs, reward, truncated, S(stream)));
    return QD_OK;
  }
  if (qd_fragment_is_persistent(env)) {
    QD_HIP(launch_rollout_coop(k, env->spec, T, actions, obs, reward, truncated, S(stream)));
    return QD_OK;
  }
  if (qd_fragment_is_pair(env)) {
    QD_HIP(launch_rollout_pair(k, T, actions, obs, reward, truncated, S(stream)));
    return QD_OK;
  }
  if (env->load) {
    // The load model outside the persistent kernels (several substeps per step, or QD_OPT_PERSISTENT_FRAGMENTS = 0): launch by
    // launch.  The one-wavefront multi-step kernel that used to run here kept the whole model and state alive across the float64
    // core -- 400-418 registers, every second one an AGPR copy, 4-34 spills (profiles/r03_kernel_resources.txt) -- for
    // configurations no training script of the reference uses (skip_steps = 1 throughout); it was removed in round 4.
    const size_t D = (size_t)env->D;
    for (int t = 0; t < T; t++) {
      const int rc = qd_step(env, actions + (size_t)t * k.n * 4, (int64_t)k.n * 4, obs + (size_t)t * k.n * D, reward + (size_t)t * k.n,
                             truncated + (size_t)t * k.n, stream);
      if (rc != QD_OK) return rc;
    }
    return QD_OK;
  }
  const dim3 grid(blocks64(k.n)), block(64);
#define QD_ROLL(SPECV) QD_LAUNCH((k_rollout<false, 64, SPECV>), grid, block, 0, S(stream), k, T, actions, obs, reward, truncated)
  if (env->spec == SPEC_SIMPLE) QD_ROLL(SPEC_SIMPLE);
  else if (env->spec == SPEC_GENERIC_FS1) QD_ROLL(SPEC_GENERIC_FS1);
  else QD_ROLL(SPEC_GENERIC);
#undef QD_ROLL
  QD_LAUNCH_CHECK();
  return QD_OK;
}

int qd_pid_reset(qd_env* env, const uint8_t* mask, void* stream) {
  QD_NEED(env);
  QD_LAUNCH(k_pid_reset, dim3(blocks64(env->ka.n)), dim3(64), 0, S(stream), env->ka, mask);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

int qd_pid_action(qd_env* env, float* actions, void* stream) {
  QD_NEED(env);
  if (!actions) return fail(QD_ERR_INVALID, "null output");
  if (env->ka.term_kind == QD_TERM_SIMPLE) return fail(QD_ERR_UNSUPPORTED, "the PID cascade drives BaseDroneEnv models (attitude_test.py), not SimpleDrone");
  const dim3 grid(blocks64(env->ka.n)), block(64);
  if (env->load) QD_LAUNCH(k_pid_action<true>, grid, block, 0, S(stream), env->ka, actions);
  else QD_LAUNCH(k_pid_action<false>, grid, block, 0, S(stream), env->ka, actions);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

int qd_rollout_pid(qd_env* env, int T, float* obs, float* reward, uint8_t* truncated, float* actions_out, void* stream) {
  QD_NEED(env);
  const KArgs& k = env->ka;
  if (T < 0) return fail(QD_ERR_INVALID, "negative step count");
  if (T == 0) return QD_OK;
  if (!obs || !reward || !truncated) return fail(QD_ERR_INVALID, "null array argument");
  if (env->ka.term_kind == QD_TERM_SIMPLE) return fail(QD_ERR_UNSUPPORTED, "the PID cascade drives BaseDroneEnv models (attitude_test.py), not SimpleDrone");
  if (env->spec == SPEC_FLOOR || (env->load && !qd_fragment_is_persistent(env))) {
    // floor contact, and the load model outside the persistent kernel (see qd_rollout): launch by launch -- controller, step, and
    // fresh controllers for the envs that were re-sampled
    if (!actions_out) return fail(QD_ERR_INVALID, "this PID rollout runs launch by launch and needs actions_out [T,N,4] as its action buffer");
    const size_t D = (size_t)env->D;
    for (int t = 0; t < T; t++) {
      float* a_t = actions_out + (size_t)t * k.n * 4;
      int rc = qd_pid_action(env, a_t, stream);
      if (rc == QD_OK) rc = qd_step(env, a_t, (int64_t)k.n * 4, obs + (size_t)t * k.n * D, reward + (size_t)t * k.n, truncated + (size_t)t * k.n, stream);
      if (rc == QD_OK && k.auto_reset) rc = qd_pid_reset(env, truncated + (size_t)t * k.n, stream);
      if (rc != QD_OK) return rc;
    }
    return QD_OK;
  }
  if (qd_fragment_is_persistent(env)) {   // the controller rides in the epilogue wave of the persistent fragment kernel
    QD_HIP(launch_rollout_coop(k, env->spec, T, nullptr, obs, reward, truncated, S(stream), true, actions_out));
    return QD_OK;
  }
  const dim3 grid(blocks64(k.n)), block(64);   // the single-body model: one wavefront per 64 envs, controller and step in one loop
#define QD_ROLL(SPECV) QD_LAUNCH((k_rollout_pid<false, SPECV>), grid, block, 0, S(stream), k, T, obs, reward, truncated, actions_out)
  if (env->spec == SPEC_GENERIC_FS1) QD_ROLL(SPEC_GENERIC_FS1);
  else QD_ROLL(SPEC_GENERIC);
#undef QD_ROLL
  QD_LAUNCH_CHECK();
  return QD_OK;
}

#include "qd_policy_host.inc"

int qd_observe(qd_env* env, float* obs, void* stream) {
  QD_NEED(env);
  if (!obs) return fail(QD_ERR_INVALID, "null output");
  return launch_observe(env, obs, stream);
}

int qd_drone_states(qd_env* env, float* states, void* stream) {
  QD_NEED(env);
  if (!states) return fail(QD_ERR_INVALID, "null output");
  QD_BY_MODEL(env, k_drone_states, dim3(blocks64(env->ka.n)), dim3(64), stream, env->ka, states);
  return QD_OK;
}

static int eval_args(EvalArgs* a, int ns, const double ref[4], int n) {
  if (ns != 33 && ns != 29) return fail(QD_ERR_INVALID, "state vectors have 33 or 29 entries, got %d", ns);
  if (!ref) return fail(QD_ERR_INVALID, "null reference");
  if (n < 0) return fail(QD_ERR_INVALID, "negative row count");
  for (int i = 0; i < 4; i++) a->ref[i] = (float)ref[i];
  a->n = n;
  return QD_OK;
}

int qd_eval_obs(int obs_kind, int ns, const float* states, const double ref_host[4], float* obs, int n, void* stream) {
  EvalArgs a{};
  int rc = eval_args(&a, ns, ref_host, n);
  if (rc) return rc;
  if (obs_kind < 0 || obs_kind >= QD_OBS_SIMPLE) return fail(QD_ERR_INVALID, "unknown observation variant %d", obs_kind);
  if (obs_kind == QD_OBS_PRY_ACC_PARAMS_NOPEND) return fail(QD_ERR_UNSUPPORTED, "variant raises NameError in the reference");
  if (n == 0) return QD_OK;
  if (!states || !obs) return fail(QD_ERR_INVALID, "null array argument");
  a.kind = obs_kind;
  a.D = obs_dim(obs_kind, ns);
  if (ns == 33) QD_LAUNCH((k_eval_obs<33>), dim3(blocks64(n)), dim3(64), 0, S(stream), a, states, obs);
  else QD_LAUNCH((k_eval_obs<29>), dim3(blocks64(n)), dim3(64), 0, S(stream), a, states, obs);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

int qd_eval_reward(int reward_kind, int ns, const float* states, const float* actions, const int32_t* num_steps,
                   const double ref_host[4], double max_distance, float* reward, int n, void* stream) {
  EvalArgs a{};
  int rc = eval_args(&a, ns, ref_host, n);
  if (rc) return rc;
  if (reward_kind < 0 || reward_kind >= QD_REW_COUNT) return fail(QD_ERR_INVALID, "unknown reward %d", reward_kind);
  if (ns == 29 && needs_load_reward(reward_kind)) return fail(QD_ERR_UNSUPPORTED, "reward needs the 33-element state");
  if (n == 0) return QD_OK;
  if (!states || !actions || !reward) return fail(QD_ERR_INVALID, "null array argument");
  a.kind = reward_kind;
  a.max_distance = (float)max_distance;
  if (ns == 33) QD_LAUNCH((k_eval_reward<33>), dim3(blocks64(n)), dim3(64), 0, S(stream), a, states, actions, num_steps, reward, (uint8_t*)nullptr);
  else QD_LAUNCH((k_eval_reward<29>), dim3(blocks64(n)), dim3(64), 0, S(stream), a, states, actions, num_steps, reward, (uint8_t*)nullptr);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

int qd_eval_truncated(int ns, const float* states, const int32_t* num_steps, const double ref_host[4], double max_distance,
                      int max_steps, uint8_t* truncated, int n, void* stream) {
  EvalArgs a{};
  int rc = eval_args(&a, ns, ref_host, n);
  if (rc) return rc;
  if (n == 0) return QD_OK;
  if (!states || !truncated) return fail(QD_ERR_INVALID, "null array argument");
  a.max_distance = (float)max_distance;
  a.max_steps = max_steps;
  if (ns == 33) QD_LAUNCH((k_eval_reward<33>), dim3(blocks64(n)), dim3(64), 0, S(stream), a, states, (const float*)nullptr, num_steps, (float*)nullptr, truncated);
  else QD_LAUNCH((k_eval_reward<29>), dim3(blocks64(n)), dim3(64), 0, S(stream), a, states, (const float*)nullptr, num_steps, (float*)nullptr, truncated);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

int qd_transform(int which, const float* in, float* out, int n, void* stream) {
  if (which < QD_TF_QUAT2RPY || which > QD_TF_PENDRP2QUAT) return fail(QD_ERR_INVALID, "unknown transform %d", which);
  if (n < 0) return fail(QD_ERR_INVALID, "negative row count");
  if (n == 0) return QD_OK;
  if (!in || !out) return fail(QD_ERR_INVALID, "null array argument");
  QD_LAUNCH(k_transform, dim3(blocks64(n)), dim3(64), 0, S(stream), which, in, out, n);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

// ---- what the reference logs about a train batch (custom_logging.py:9-31, training.py:16-22), computed where the fragments lie ----
size_t qd_column_stats_workspace_bytes(int cols) {
  if (cols < 1 || cols > qd::STAT_MAX_COLS) return 0;
  return (size_t)qd::STAT_GROUPS * cols * 4 * sizeof(double);
}

int qd_column_stats(const float* x, int64_t rows, int cols, double* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (cols < 1 || cols > qd::STAT_MAX_COLS) return fail(QD_ERR_UNSUPPORTED, "column statistics support 1..%d columns, got %d", qd::STAT_MAX_COLS, cols);
  if (rows < 1) return fail(QD_ERR_INVALID, "column statistics of an empty batch are undefined (numpy raises / warns)");
  if (!x || !out || !workspace) return fail(QD_ERR_INVALID, "null array argument");
  if (workspace_bytes < qd_column_stats_workspace_bytes(cols) || (reinterpret_cast<uintptr_t>(workspace) & 7))
    return fail(QD_ERR_ARENA, "workspace too small (qd_column_stats_workspace_bytes) or not 8-byte aligned");
  if (reinterpret_cast<uintptr_t>(x) & 3) return fail(QD_ERR_INVALID, "matrix not 4-byte aligned");
  // the pass reads 16-byte units; the floats before the first boundary and after the last whole unit go to the final kernel
  const long long total = (long long)rows * cols;
  int head = (int)(((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / 4);
  if (head > total) head = (int)total;
  const long long units = (total - head) / 4;
  const long long span = (long long)qd::stat_period(cols) * qd::STAT_UNROLL;
  long long groups = (units + span - 1) / span;
  if (groups < 1) groups = 1;
  if (groups > qd::STAT_GROUPS) groups = qd::STAT_GROUPS;
  double* part = static_cast<double*>(workspace);
  QD_LAUNCH(qd::k_column_stats, dim3((unsigned)groups), dim3(qd::STAT_THREADS), 0, S(stream), reinterpret_cast<const qd::stat_f4*>(x + head), units, cols,
            head, part);
  QD_LAUNCH_CHECK();
  QD_LAUNCH(qd::k_column_stats_final, dim3((cols + qd::STAT_FINAL_COLS - 1) / qd::STAT_FINAL_COLS), dim3(qd::STAT_FINAL_THREADS), 0, S(stream), part,
            (int)groups, cols, (long long)rows, x, head, units, out);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

// workspace: [segment summaries: returns S*N*2 doubles | lengths S*N*2 ints | partials (S+1)*G*8 doubles], sized for the most segments
static void epi_layout(int num_envs, int S, size_t* ret_off, size_t* len_off, size_t* part_off, size_t* total) {
  const size_t G = (size_t)(num_envs + qd::STAT_THREADS - 1) / qd::STAT_THREADS;
  *ret_off = 0;
  *len_off = (size_t)S * num_envs * 2 * sizeof(double);
  *part_off = (*len_off + (size_t)S * num_envs * 2 * sizeof(int) + 15) / 16 * 16;
  *total = *part_off + (size_t)(S + 1) * G * qd::EPI_FIELDS * sizeof(double);
}

// time segments per fragment: enough (segment, env-group) workgroups to fill the chip a few times over -- 64 segments for the
// BASELINE batch sizes, fewer when the env axis alone supplies the parallelism (the summaries cost 24 bytes per segment and env)
static int epi_max_segments(int num_envs) {
  const int G = (num_envs + qd::STAT_THREADS - 1) / qd::STAT_THREADS;
  int s = 4096 / G;
  if (s > qd::EPI_MAX_SEGMENTS) s = qd::EPI_MAX_SEGMENTS;
  return s < 1 ? 1 : s;
}

size_t qd_episode_stats_workspace_bytes(int num_envs) {
  if (num_envs < 1) return 0;
  size_t a, b, c, total;
  epi_layout(num_envs, epi_max_segments(num_envs), &a, &b, &c, &total);
  return total;
}

int qd_episode_stats(const float* reward, const uint8_t* truncated, int T, int num_envs, double* carry, double* out, void* workspace,
                     size_t workspace_bytes, void* stream) {
  if (T < 0 || num_envs < 1) return fail(QD_ERR_INVALID, "bad fragment shape [%d, %d]", T, num_envs);
  if (!reward || !truncated || !carry || !out || !workspace) return fail(QD_ERR_INVALID, "null array argument");
  if (workspace_bytes < qd_episode_stats_workspace_bytes(num_envs) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return fail(QD_ERR_ARENA, "workspace too small (qd_episode_stats_workspace_bytes) or not 16-byte aligned");
  const int G = (num_envs + qd::STAT_THREADS - 1) / qd::STAT_THREADS;
  // short walks: as many segments as the workspace is sized for (epi_max_segments), none shorter than 8 steps
  int nseg = epi_max_segments(num_envs);
  if (nseg > T / qd::EPI_MIN_SEGMENT) nseg = T / qd::EPI_MIN_SEGMENT;
  if (nseg < 1) nseg = 1;
  const int L = T > 0 ? (T + nseg - 1) / nseg : 1;
  nseg = T > 0 ? (T + L - 1) / L : 1;
  size_t ret_off, len_off, part_off, total;
  epi_layout(num_envs, nseg, &ret_off, &len_off, &part_off, &total);
  char* ws = static_cast<char*>(workspace);
  double* seg_ret = reinterpret_cast<double*>(ws + ret_off);
  int* seg_len = reinterpret_cast<int*>(ws + len_off);
  double* part = reinterpret_cast<double*>(ws + part_off);
  QD_LAUNCH(qd::k_episode_segments, dim3(G, nseg), dim3(qd::STAT_THREADS), 0, S(stream), reward, truncated, T, num_envs, L, seg_ret, seg_len, part);
  QD_LAUNCH_CHECK();
  QD_LAUNCH(qd::k_episode_stitch, dim3(G), dim3(qd::STAT_THREADS), 0, S(stream), seg_ret, seg_len, nseg, num_envs, carry, part,
            part + (size_t)nseg * G * qd::EPI_FIELDS);
  QD_LAUNCH_CHECK();
  QD_LAUNCH(qd::k_episode_stats_final, dim3(1), dim3(qd::STAT_THREADS), 0, S(stream), part + (size_t)nseg * G * qd::EPI_FIELDS, G, out);
  QD_LAUNCH_CHECK();
  return QD_OK;
}

}  // extern "C"
