// qd_rollout_lat.hip -- k_rollout_lat: the persistent fragment kernel of the training configuration for batches that leave every
// workgroup a CU to itself (<= 256 workgroups = 16384 envs; BASELINE config 3 / 4 at 4096 envs per GPU).
//
// Replaces, for T consecutive calls, BaseDroneEnv.vector_step (environments/BaseDroneEnv.py:259-294) as the sampler of
// train_PPO.py / train_RMA.py drives it in 1024-step fragments (train_RMA.py:63) -- like k_rollout_coop (qd_rollout_coop.hip),
// whose reset pool and contract it keeps.  What differs is the cut of the work.  At these sizes a step is a latency chain on
// four SIMDs, and measured on MI355X (tools/microbench/lane_skip.hip) a wavefront issues one vector instruction per 4.2-5 cycles
// whether or not it depends on the previous one, float32 or float64: the length of a phase IS the instruction count of its
// longest wave plus the LDS round trips at its ends.  k_rollout_coop's phase 1 was as long as the solver wave's factorisation
// (421 instructions) while the epilogue wave spent the same phase building a row nobody waits for, and its phase 2 ran three
// waves idle beside the solve.  Here:
//
//   wave A (solver)    mass_inverse(s_t): inertia from per-env      | solve_inv5 (5 dot products), integration published plane by
//                      coefficients, hinges first, adjugate;        | plane as it becomes final, truncation, reset from the pool
//                      attitude matrix, normalised quaternion       | entry PREFETCHED IN REGISTERS, the quaternion's chain last
//   wave B (airframe)  motor filter, attitude, rotors, drag on core | the observation row of s_t (= row of step t - 1) into an LDS
//                      and link                                     | tile [SPEC_LSTM: and the stores of row t - 2]
//   wave C (inertial)  inertial_wrench(s_t)                         | one chunk of the workgroup's reset sampler
//   wave D (tether)    attitude, tether geometry, drag on the tether| the reward of step t - 1, streaming stores of row t - 2
//                    barrier 1 ^                                                                          barrier 2 ^
//
// so that phase 1 carries only what the solve waits for, split four ways, and the epilogue lives in phase 2 beside the solve.
// The arithmetic of the mass matrix is qd_dynamics.h's latency arrangement (lat_consts / mass_inverse / solve_inv5: same
// equations as mass_factor / reduce_rhs / finish_accel, tests/test_host_twin.py).  Rows that carry the accelerometer get the
// damping-explicit accelerations from the implicit solve by a 2 x 2 correction (explicit_from_implicit) and are handed over a round
// late (L.accv; in train_LSTM.py's configuration, SPEC_LSTM, the row wave also stores the rows: FLUSH_B).  The PID action source
// stays with k_rollout_coop, as do batches of more than 256 workgroups (there the SIMDs' issue slots are the bound and this
// kernel's duplicated attitude / geometry arithmetic would cost throughput).
#include "qd_env_device.h"

namespace qd {

constexpr int RL_THREADS = 256;

// The state as wave A publishes it, grouped by WHEN a value is final inside the solver's phase 2, so that the LDS writes leave while
// the rest of the integration still runs (a block of six writes behind the last value cost ~170 cycles of drain in front of barrier 2):
// velocities and hinges right behind the solve, the position one multiply-add later, the quaternion -- a 60-instruction chain -- last.
// The activations are not in it: the motors' filter belongs to wave B (the only wave that reads them).
enum { RL_V = 0, RL_W, RL_H, RL_P, RL_Q, RL_PLANES };   // (vx,vy,vz,-) (wx,wy,wz,-) (th1,th2,thd1,thd2) (px,py,pz,-) (qw,qx,qy,qz)

constexpr int RL_TAG = RL_PLANES;   // the pool's tag plane: (-, episode, stage, -)
struct RlLds {
  float4 appB[2][64];         // B -> A: (F, t1) (Tq, -) of rotors + drag on core and link, minus the core body's inertial share
  float4 appD[2][64];         // D -> A: (F, t1) (Tq, t2) of the drag on the tether
  double2 ine[4][64];         // C -> A: Inertial (F, Tq, t1, t2) of link and tether
  float4 st[RL_PLANES][64];   // A -> B, C, D: the state AFTER the step, before any reset (the reward of a truncated lane is of this state)
  uint4 info[64];             // A -> B, C, D: (bit 0 truncated | bit 1 reset), episode counter of s_{t+1}, num_steps after the step, -
  float4 accv[64];            // A -> B, D (sensor-reading rows): the accelerometer reading of the step solved in the last phase 2
  // The reset pool: slot e & 1 = entry of episode e, in the planes of `st` plus the tag plane.  A lane that is reset takes its new state
  // from here -- EVERY wave for itself, at the start of the next phase 1 (rl_get_start): the entry is in LDS already, so a reset costs the
  // solver wave no write at all (writing the pre-reset and the new state out for the other waves: +740 cycles on its critical phase in
  // each of the 36 % of steps with a resetting lane, profiles/r04_rollout_lat_notes.txt)
  float4 nxt[2][RL_PLANES + 1][64];
  float tile[2][64 * QD_MAX_OBS];   // observation rows, row-major like the global span: wave B fills one while wave D streams the other out
};

__device__ __forceinline__ void rl_put_all(float4 (*st)[64], int lane, const State<float>& s) {
  st[RL_V][lane] = make_float4(s.vx, s.vy, s.vz, 0.f);
  st[RL_W][lane] = make_float4(s.wx, s.wy, s.wz, 0.f);
  st[RL_H][lane] = make_float4(s.th1, s.th2, s.thd1, s.thd2);
  st[RL_P][lane] = make_float4(s.px, s.py, s.pz, 0.f);
  st[RL_Q][lane] = make_float4(s.qw, s.qx, s.qy, s.qz);
}
// the planes a role needs (VEL: also the linear velocity, POS: also the position)
// Returns the reset mark that travels in the spare slot of the W plane: 0 = the lane goes on from this state, 1 + s = the step that
// produced it truncated the lane and the new episode's state is the pool entry in slot s.  In a plane every role waits for anyway:
// a mark in `info` put its LDS round trip and the branch behind it at the head of the three wrench waves' phase 1 (+200 cycles).
template <bool VEL, bool POS>
__device__ __forceinline__ uint32_t rl_get(const float4 (*st)[64], int lane, State<float>& s) {
  const float4 q = st[RL_Q][lane], w = st[RL_W][lane], h = st[RL_H][lane];
  s.qw = q.x; s.qx = q.y; s.qy = q.z; s.qz = q.w;
  s.wx = w.x; s.wy = w.y; s.wz = w.z;
  s.th1 = h.x; s.th2 = h.y; s.thd1 = h.z; s.thd2 = h.w;
  if (VEL) { const float4 v = st[RL_V][lane]; s.vx = v.x; s.vy = v.y; s.vz = v.z; }
  if (POS) { const float4 p = st[RL_P][lane]; s.px = p.x; s.py = p.y; s.pz = p.z; }
  return __float_as_uint(w.w);
}
// the state a step STARTS from: what wave A published, and for the lanes it reset (a wave-uniform branch: one step in three has such a
// lane among its 64) the pool entry of the episode that begins (`mark` from rl_get)
template <bool VEL, bool POS>
__device__ __forceinline__ void rl_take_reset(const RlLds& L, int lane, uint32_t mark, State<float>& s) {
  const bool rst = mark != 0u;
  if (__any(rst ? 1 : 0)) {
    State<float> r;
    rl_get<VEL, POS>(L.nxt[(mark - 1u) & 1u], lane, r);
    if (rst) {
      s.qw = r.qw; s.qx = r.qx; s.qy = r.qy; s.qz = r.qz; s.wx = r.wx; s.wy = r.wy; s.wz = r.wz;
      s.th1 = r.th1; s.th2 = r.th2; s.thd1 = r.thd1; s.thd2 = r.thd2;
      if (VEL) { s.vx = r.vx; s.vy = r.vy; s.vz = r.vz; }
      if (POS) { s.px = r.px; s.py = r.py; s.pz = r.pz; }
    }
  }
}
__device__ __forceinline__ void rl_pool_put(float4 (*slot)[64], int lane, uint32_t episode, const State<float>& s) {
  rl_put_all(slot, lane, s);
  slot[RL_TAG][lane] = make_float4(0.f, __uint_as_float(episode), __uint_as_float(POOL_STATE), 0.f);
}
__device__ __forceinline__ bool rl_entry_valid(float4 tagp, uint32_t episode) {
  return __float_as_uint(tagp.z) != 0u && __float_as_uint(tagp.y) == episode;
}
// arena planes (pool_store's layout) <-> the LDS pool's
__device__ __forceinline__ void rl_pool_load(const KArgs& a, int base, int il, float4 (*slot)[64], int lane, bool on) {
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f), q = p, v = p, w = p, x = p;
  if (on) {
    p = a.g[(base + 0) * a.npad + il]; q = a.g[(base + 1) * a.npad + il]; v = a.g[(base + 2) * a.npad + il];
    w = a.g[(base + 3) * a.npad + il]; x = a.g[(base + 4) * a.npad + il];
  }
  slot[RL_V][lane] = make_float4(v.x, v.y, v.z, 0.f);
  slot[RL_W][lane] = make_float4(w.x, w.y, w.z, 0.f);
  slot[RL_H][lane] = make_float4(p.w, v.w, w.w, x.x);
  slot[RL_P][lane] = make_float4(p.x, p.y, p.z, 0.f);
  slot[RL_Q][lane] = q;
  slot[RL_TAG][lane] = make_float4(0.f, x.y, x.z, 0.f);
}
__device__ __forceinline__ void rl_pool_store(const KArgs& a, int base, int i, const float4 (*slot)[64], int lane) {
  const float4 v = slot[RL_V][lane], w = slot[RL_W][lane], h = slot[RL_H][lane], p = slot[RL_P][lane], q = slot[RL_Q][lane], x = slot[RL_TAG][lane];
  a.g[(base + 0) * a.npad + i] = make_float4(p.x, p.y, p.z, h.x);
  a.g[(base + 1) * a.npad + i] = q;
  a.g[(base + 2) * a.npad + i] = make_float4(v.x, v.y, v.z, h.y);
  a.g[(base + 3) * a.npad + i] = make_float4(w.x, w.y, w.z, h.z);
  a.g[(base + 4) * a.npad + i] = make_float4(h.w, x.y, x.z, 0.f);
}
// flush_obs_static (qd_env_device.h) in its two halves: all LDS reads of the wave's 64 rows, and later their streaming stores
template <int D>
__device__ __forceinline__ void rl_flush_load(const float* tile, float4* v, float4& vt) {
  constexpr int N4 = 16 * D, FULL = N4 / 64, TAIL = N4 % 64;
  const int lane = threadIdx.x & 63;
  const float4* t4 = reinterpret_cast<const float4*>(tile) + lane;
#pragma unroll
  for (int k = 0; k < FULL; k++) v[k] = t4[64 * k];
  if (TAIL > 0 && lane < TAIL) vt = t4[64 * FULL];
}
template <int D>
__device__ __forceinline__ void rl_flush_store(float* dst, const float4* v, const float4& vt) {
  constexpr int N4 = 16 * D, FULL = N4 / 64, TAIL = N4 % 64;
  const int lane = threadIdx.x & 63;
  float4* d4 = reinterpret_cast<float4*>(dst) + lane;
#pragma unroll
  for (int k = 0; k < FULL; k++) store_streaming(d4 + 64 * k, v[k]);
  if (TAIL > 0 && lane < TAIL) store_streaming(d4 + 64 * FULL, vt);
}
__device__ __forceinline__ void rl_pin(double x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void rl_pin(const V3<double>& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z)); }
__device__ __forceinline__ void rl_pin(float x, float y, float z, float w) { asm volatile("" ::"v"(x), "v"(y), "v"(z), "v"(w)); }

#ifdef QD_STAMPS
// diagnostic build: cycle stamps of one step in the middle of the fragment, 16 per (workgroup, wave)
__device__ unsigned long long qd_rlstamps[64 * 4 * 16];
#define RL_STAMP(k)                                                                                          \
  do {                                                                                                       \
    if (t == (T >> 1)) {                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      unsigned long long t_;                                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      if (lane == 0 && blockIdx.x < 64) qd_rlstamps[(blockIdx.x * 4 + role) * 16 + (k)] = t_;                \
    }                                                                                                        \
  } while (0)
#else
#define RL_STAMP(k)
#endif

// SPEC_RMA (train_PPO.py / train_RMA.py) or SPEC_GENERIC_FS1 with an observation variant that does not read the accelerometer
// (any reward; dispatched at run time in wave B); one substep per step.
template <int SPEC>
__global__ __launch_bounds__(RL_THREADS, 1) void k_rollout_lat(KArgs a, int T, const float* __restrict__ actions, float* __restrict__ obs,
                                                               float* __restrict__ reward_out, uint8_t* __restrict__ trunc_out) {
  static_assert(SPEC == SPEC_RMA || SPEC == SPEC_LSTM || SPEC == SPEC_GENERIC_FS1, "latency-bound fragment kernel: the load model, one substep per step");
  const int D = spec_runtime<SPEC>() ? a.D : spec_obs_dim<SPEC>();
  // Observation variants that carry the accelerometer (`sens`: train_LSTM.py's LocalFrameFullStateEnv, BaseDroneEnv's raw row, the
  // ...acc... wrappers).  The reading in the row of step k is the one mj_step computed in that step at the state it STARTED from
  // (quirk C-6): the damping-explicit accelerations of round k's solve, which the solver wave gets from its implicit solution by a
  // 2 x 2 correction (explicit_from_implicit) and publishes in L.accv.  The row of a lane that was reset in step k instead carries
  // mj_forward's reading at the NEW state with the activations that survived (set_state -> mj_forward, mujoco_vecenv.py:396-402):
  // round k + 1's reading -- the row wave leaves those three entries open and the store wave fills them in a round later, before
  // the row goes out.  The fragment's last row needs the reading at s_T: one extra half round (wrenches and solve, nothing integrated).
  const bool sens = SPEC == SPEC_RMA ? false : (SPEC == SPEC_LSTM ? true : a.obs_needs_acc != 0);
  // Who streams the finished rows out: wave D, behind its reward -- except in train_LSTM.py's configuration, whose pendulum-energy
  // reward (rewards.py:191-230: ~150 instructions more than distance_energy_reward) makes wave D the last wave of phase 2 by 470
  // cycles; there the row wave, which has the slack, stores the rows it built a round ago (config 5 at 8192 envs: 1.61 -> 1.4x us)
  constexpr bool FLUSH_B = SPEC == SPEC_LSTM;
  const int acc_at = sens ? rc_acc_slot(spec_obs<SPEC>(a)) : -1;
  __shared__ RlLds L;
  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int base_env = blockIdx.x * 64;
  const int i = base_env + lane;
  const bool live = i < a.n;
  const int il = live ? i : a.n - 1;   // lanes past the batch work on a copy of the last env (no stores, no atomics)
  const int n = a.n;
  const float4* actions4 = reinterpret_cast<const float4*>(actions);

  // every wave fetches its env's planes (what a role does not read is dropped by the compiler); the reference once
  EnvRegs e;
  load_env_planes<true, false, false>(a.g, a.npad, il, e);
  float ref0[4] = {a.ref[0], a.ref[1], a.ref[2], a.ref[3]};
  if (a.ref_mode == QD_REF_STATIC && a.per_env_ref) {
    const float4 r = a.g[G_REF * a.npad + il];
    ref0[0] = r.x; ref0[1] = r.y; ref0[2] = r.z; ref0[3] = r.w;
  }

  if (role == 0) {
    // ================================================================ wave A: inverse of the reduced mass matrix, solve, integration, resets
    rl_put_all(L.st, lane, e.s);
    L.info[lane] = make_uint4(0u, e.episode, (uint32_t)e.num_steps, 0u);
    const LatConsts<double> K = lat_consts(e.M, a.h);
    coop_barrier();   // P
    // does the pool hold the entry of this env's CURRENT episode counter?  (looked up behind the last publication of a phase 2, where the
    // round trip hides behind the barrier; a reset then only advances counters here -- every wave fetches the entry for itself)
    bool have = a.use_pool ? rl_entry_valid(L.nxt[e.episode & 1u][RL_TAG][lane], e.episode) : false;
    uint32_t mark_prev = 0u;
    // what the accelerometer reading of the LAST step needs (it is left in the arena like a per-step launch leaves it): that step's
    // two wrenches, attitude, body rates and hinge angles; evaluated once, behind the loop
    Applied<float> ap_last;
    ap_last.F = ap_last.Tq = mk<float>(0.f, 0.f, 0.f); ap_last.t1 = ap_last.t2 = 0.f;
    Inertial<double> in_last;
    in_last.F = in_last.Tq = mk<double>(0.0, 0.0, 0.0); in_last.t1 = in_last.t2 = 0.0;
    M3<float> R;
    R.m00 = R.m11 = R.m22 = 1.f; R.m01 = R.m02 = R.m10 = R.m12 = R.m20 = R.m21 = 0.f;
    V3<float> w0 = mk<float>(0.f, 0.f, 0.f);
    float th1_last = e.s.th1, th2_last = e.s.th2;
    V3<float> acc_step = mk<float>(0.f, 0.f, 0.f), acc_before = mk<float>(0.f, 0.f, 0.f);   // the readings of the last two solves
    for (int t = 0; t < T + (sens ? 1 : 0); t++) {
      RL_STAMP(0);
      const bool half = t == T;   // `sens` only: the forward dynamics at s_T, nothing integrated
      // ---------------------------------------------------------- phase 1
      rl_take_reset<true, true>(L, lane, mark_prev, e.s);   // this wave reset the lane a phase ago: the new episode's state
      th1_last = e.s.th1; th2_last = e.s.th2;
      float s1, c1, s2, c2;
      qsincos(e.s.th1, &s1, &c1);
      qsincos(e.s.th2, &s2, &c2);
      const TetherHP<double> th = tether_hp<double>(s1, c1, s2, c2);
      const Inv5<double> v5 = mass_inverse(K, th);
      // the attitude of s_t: its matrix (for the linear acceleration) and the normalised quaternion the exponential map starts from
      const float qn = frsq(e.s.qw * e.s.qw + e.s.qx * e.s.qx + e.s.qy * e.s.qy + e.s.qz * e.s.qz);
      const float qhw = e.s.qw * qn, qhx = e.s.qx * qn, qhy = e.s.qy * qn, qhz = e.s.qz * qn;
      R = quat2mat(qhw, qhx, qhy, qhz);
      rc_ref(a, i, e.num_steps, ref0, e.ref);
      // everything the solve reads exists BEFORE the barrier (the barrier is an asm the compiler moves pure arithmetic across)
      rl_pin(v5.cxx); rl_pin(v5.cxy); rl_pin(v5.cxz); rl_pin(v5.cyy); rl_pin(v5.cyz); rl_pin(v5.czz);
      rl_pin(v5.U1); rl_pin(v5.U2); rl_pin(v5.s11); rl_pin(v5.s12); rl_pin(v5.s22);
      rl_pin(v5.rc); rl_pin(v5.kp1.y); rl_pin(v5.kp1.z); rl_pin(v5.kp2);
      rl_pin(qhw, qhx, qhy, qhz); rl_pin(R.m00, R.m01, R.m02, R.m10); rl_pin(R.m11, R.m12, R.m20, R.m21); rl_pin(R.m22, e.ref[0], e.ref[1], e.ref[2]);
      RL_STAMP(1);
      coop_barrier();   // 1
      RL_STAMP(2);
      // ---------------------------------------------------------- phase 2
      Applied<float> ap;
      Inertial<double> in;
      {
        const float4 b0 = L.appB[0][lane], b1 = L.appB[1][lane], d0 = L.appD[0][lane], d1 = L.appD[1][lane];
        const double2 y0 = L.ine[0][lane], y1 = L.ine[1][lane], y2 = L.ine[2][lane], y3 = L.ine[3][lane];
        ap.F = mk<float>(b0.x + d0.x, b0.y + d0.y, b0.z + d0.z); ap.t1 = b0.w + d0.w;
        ap.Tq = mk<float>(b1.x + d1.x, b1.y + d1.y, b1.z + d1.z); ap.t2 = d1.w;
        in.F = mk<double>(y0.x, y0.y, y1.x); in.Tq = mk<double>(y1.y, y2.x, y2.y); in.t1 = y3.x; in.t2 = y3.y;
      }
      if (!half) { ap_last = ap; in_last = in; }
      w0 = mk<float>(e.s.wx, e.s.wy, e.s.wz);
      const float h = a.h;
      const Rot5<double> r5 = solve_inv5_rot(v5, ap, in);
      if (sens) {   // the accelerometer reading of this solve: explicit accelerations by the 2 x 2 correction of the implicit ones
        V3<double> a0ex;
        V3<float> angex;
        float d1, d2;
        explicit_from_implicit(K, v5, explicit_weights(K, v5), r5.fl, r5.al, r5.t1, r5.t2, &a0ex, &angex, &d1, &d2);
        const float g = float(Const::gravity);
        acc_before = acc_step;
        acc_step = accelerometer(cvt<float>(a0ex), angex, mk<float>(g * R.m20, g * R.m21, g * R.m22),
                                 mk<float>(w0.x * w0.z, w0.y * w0.z, -(w0.x * w0.x + w0.y * w0.y)));
        L.accv[lane] = make_float4(acc_step.x, acc_step.y, acc_step.z, 0.f);
      }
      if (half) {
        RL_STAMP(3);
        coop_barrier();   // 2 (of the half round: the reading at s_T is published)
        RL_STAMP(4);
        break;
      }
      {
        const V3<float> ang = cvt<float>(r5.al);
        const float thdd1 = (float)r5.t1, thdd2 = (float)r5.t2;
        const V3<float> lin = mul(R, cvt<float>(solve_inv5_lin(K, v5, r5)));
        // integrate_motion()'s arithmetic, published as it becomes final
        e.s.vx += h * lin.x; e.s.vy += h * lin.y; e.s.vz += h * lin.z;
        e.s.wx += h * ang.x; e.s.wy += h * ang.y; e.s.wz += h * ang.z;
        e.s.thd1 += h * thdd1; e.s.thd2 += h * thdd2;
        e.s.th1 += h * e.s.thd1; e.s.th2 += h * e.s.thd2;
        L.st[RL_V][lane] = make_float4(e.s.vx, e.s.vy, e.s.vz, 0.f);
        L.st[RL_H][lane] = make_float4(e.s.th1, e.s.th2, e.s.thd1, e.s.thd2);
        e.s.px += h * e.s.vx; e.s.py += h * e.s.vy; e.s.pz += h * e.s.vz;
        L.st[RL_P][lane] = make_float4(e.s.px, e.s.py, e.s.pz, 0.f);
      }
      {  // the attitude: the longest chain of the integration, beside the linear one (truncation test) in the same basic block
        float nw, nx, ny, nz;
        quat_advance(qhw, qhx, qhy, qhz, e.s.wx, e.s.wy, e.s.wz, h, &nw, &nx, &ny, &nz);
        e.s.qw = nw; e.s.qx = nx; e.s.qy = ny; e.s.qz = nz;
        L.st[RL_Q][lane] = make_float4(nw, nx, ny, nz);
      }
      e.flags &= ~FLAG_ACC_STALE;
      e.num_steps += 1;
      const int steps_post = e.num_steps;
      bool tr;
      {  // default_termination_fcn / SimpleDrone's rule on the position alone
        const float dx = e.s.px - e.ref[0], dy = e.s.py - e.ref[1], dz = e.s.pz - e.ref[2];
        const float dist = qsqrt(dx * dx + dy * dy + dz * dz);
        tr = spec_term<SPEC>(a) == QD_TERM_SIMPLE ? dist > 0.5f : (!(dist <= a.max_distance) || e.num_steps >= a.max_steps);
      }
      const bool rst = a.auto_reset && tr;
      mark_prev = rst ? 1u + (e.episode & 1u) : 0u;
      L.st[RL_W][lane] = make_float4(e.s.wx, e.s.wy, e.s.wz, __uint_as_float(mark_prev));   // the body rates, with the reset mark
      RL_STAMP(5);
      if (__any(rst ? 1 : 0)) {
        if (__any((rst && !have) ? 1 : 0)) {   // no entry (pool off, or the sampler has not got there yet): sample inline, same result,
          if (rst && !have) {                   // and leave it where every wave looks for it
            State<float> ns;
            sample_episode<true>(a, i, e.episode, ns);
            rl_pool_put(L.nxt[e.episode & 1u], lane, e.episode, ns);
          }
        }
        if (rst) {
          if (a.use_pool && live) pool_count(a, have);
          e.episode += 1u;
          e.num_steps = 0;
          // without the sensor in the row the stored reading is only marked stale (the next step, or a getter that runs first,
          // recomputes it); with it, the next round's reading takes its place
          if (!sens) e.flags |= FLAG_ACC_STALE;
          have = false;
        }
      }
      RL_STAMP(6);
      L.info[lane] = make_uint4((tr ? 1u : 0u) | (rst ? 2u : 0u), e.episode, (uint32_t)steps_post, 0u);
      // the next episode's entry: is it there by now (wave C commits at the start of a phase 1)?  Behind the last publication, so the
      // round trip hides behind the barrier; lanes that know skip it
      if (a.use_pool && !have) have = rl_entry_valid(L.nxt[e.episode & 1u][RL_TAG][lane], e.episode);
      RL_STAMP(3);
      coop_barrier();   // 2
      RL_STAMP(4);
    }
    rl_take_reset<true, true>(L, lane, mark_prev, e.s);   // (before X: behind it wave C completes the pool for the arena)
    coop_barrier();   // X: wave B's last row is in its tile
    // the fragment's last step leaves what a per-step launch leaves: the state, and the accelerometer reading of that step
    // (quirk C-6: the reading of the state the step STARTED from; where a reset invalidated it, the flag says so).  The
    // activations are wave B's to store.
    if (live) {
      if (sens) {
        // acc_step is the half round's reading (at s_T), acc_before the last step's: a lane the last step reset keeps mj_forward's
        e.acc = mark_prev != 0u ? acc_step : acc_before;
      } else {
        const Tether<float> tg = tether_geometry(th1_last, th2_last);
        const Factor<double> f = mass_factor<true>(e.M, tg, a.h);
        const Rhs<double> r = reduce_rhs<true>(f, ap_last, in_last);
        e.acc = rc_sensor(f, r, R, w0);
      }
      float4* g = a.g;
      const int np = a.npad;
      g[G_POS * np + i] = make_float4(e.s.px, e.s.py, e.s.pz, e.s.th1);
      g[G_QUAT * np + i] = make_float4(e.s.qw, e.s.qx, e.s.qy, e.s.qz);
      g[G_VEL * np + i] = make_float4(e.s.vx, e.s.vy, e.s.vz, e.s.th2);
      g[G_ANG * np + i] = make_float4(e.s.wx, e.s.wy, e.s.wz, e.s.thd1);
      g[G_AUX * np + i] = make_float4(e.s.thd2, __int_as_float(e.num_steps), __uint_as_float(e.episode), __uint_as_float(e.flags));
      g[G_ACC * np + i] = make_float4(e.acc.x, e.acc.y, e.acc.z, 0.f);
    }
  } else if (role == 1) {
    // ================================================================ wave B: motor filter, rotors, drag on core and link | rows, rewards
    float4 act_now = actions4[il];                        // u_t when round t uses it
    float4 act_prev = make_float4(0.f, 0.f, 0.f, 0.f);    // u_{t-1}: the reward of step t - 1 reads it
    coop_barrier();   // P
    const int rows_b = min(64, n - base_env);
    uint32_t markb_before = 0u;   // FLUSH_B: the lanes reset two steps ago (their row's sensor entries are due when it goes out)
    for (int t = 0; t <= T; t++) {
      RL_STAMP(0);
      EnvRegs ed;   // what write_obs_row reads of an env: its state and reference
      const uint32_t mark = rl_get<true, true>(L.st, lane, ed.s);
      const uint4 info = L.info[lane];   // of s_t (wave A rewrites it in phase 2)
      const bool rst = mark != 0u;
      rl_take_reset<true, true>(L, lane, mark, ed.s);
      // the activations survive a reset (reference quirk C-2) -- unless they diverged: MuJoCo's bad-state check would have called
      // mj_resetData, which zeroes them (reset_bookkeeping)
      if (rst && !(fabsf(e.s.a0) + fabsf(e.s.a1) + fabsf(e.s.a2) + fabsf(e.s.a3) < 1e10f)) e.s.a0 = e.s.a1 = e.s.a2 = e.s.a3 = 0.f;
      ed.s.a0 = e.s.a0; ed.s.a1 = e.s.a1; ed.s.a2 = e.s.a2; ed.s.a3 = e.s.a3;   // a_t
      // the reading of the step that led here (published in the solver wave's last phase 2, rewritten in its next): the sensor entries
      // of this round's row -- except for a lane that was reset, whose entries wait for this round's reading (wave D fills them in)
      V3<float> acc_row = mk<float>(0.f, 0.f, 0.f);
      float4 accb_due = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sens && t >= 1) {
        accb_due = L.accv[lane];
        if (!rst) acc_row = mk<float>(accb_due.x, accb_due.y, accb_due.z);
      }
      if (t < T || sens) {
        M3<float> Rb;
        V3<float> vb;
        attitude_min(ed.s, &Rb, &vb);
        float s1, c1;
        qsincos(ed.s.th1, &s1, &c1);
        const float g = float(Const::gravity);
        const Applied<float> ap = applied_core_link<true>(e.M, ed.s, mk<float>(ed.s.wx, ed.s.wy, ed.s.wz), vb, s1, c1, mk<float>(g * Rb.m20, g * Rb.m21, g * Rb.m22));
        L.appB[0][lane] = make_float4(ap.F.x, ap.F.y, ap.F.z, ap.t1);
        L.appB[1][lane] = make_float4(ap.Tq.x, ap.Tq.y, ap.Tq.z, 0.f);
        // ctrl map and activation filter: a_{t+1}, the part of the Euler step that does not wait for the accelerations
        if (t < T) rc_filter<SPEC>(a, e.M, e.s, act_now);
        RL_STAMP(1);
        coop_barrier();   // 1
        RL_STAMP(2);
      }
      if (FLUSH_B && t >= 2) {   // row t - 2, built a round ago; a lane that step t - 2 reset gets the reading of step t - 1's solve first
        if (sens && markb_before != 0u) {
          float* row = L.tile[(t - 1) & 1] + lane * D + acc_at;
          row[0] = accb_due.x; row[1] = accb_due.y; row[2] = accb_due.z;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        flush_obs_any<SPEC>(L.tile[(t - 1) & 1], obs + ((size_t)(t - 2) * n + base_env) * D, rows_b, D);
      }
      markb_before = mark;
      if (t >= 1) {   // the row of step t - 1 is the observation of s_t; its reward is of the state before a reset
        float ref_t[4];
        rc_ref(a, i, (int)info.z - 1, ref0, ref_t);   // the reference the step ran with (episode step before the increment)
        ed.ref[0] = ref_t[0]; ed.ref[1] = ref_t[1]; ed.ref[2] = ref_t[2]; ed.ref[3] = ref_t[3];
        if (rst && a.ref_mode != QD_REF_STATIC) moving_reference(a, i, 0, ed.ref);   // a new episode's first row
        float sv[33];
        M3<float> Rq;
        drone_state<float, true>(ed.s, acc_row, ed.ref, e.par, sv, &Rq);
        write_obs_row<true, SPEC>(a, ed, sv, &Rq, L.tile[t & 1] + lane * D);
      }
      act_prev = act_now;
      if (t + 1 < T) act_now = actions4[(size_t)(t + 1) * n + il];   // for round t + 1: in flight across the barrier
      RL_STAMP(3);
      if (t < T || sens) coop_barrier();   // 2
      RL_STAMP(4);
    }
    coop_barrier();   // X
    if (FLUSH_B) {   // the last row (a lane the last step reset: the half round's reading, at s_T)
      if (sens && markb_before != 0u) {
        const float4 x = L.accv[lane];
        float* row = L.tile[T & 1] + lane * D + acc_at;
        row[0] = x.x; row[1] = x.y; row[2] = x.z;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      flush_obs_any<SPEC>(L.tile[T & 1], obs + ((size_t)(T - 1) * n + base_env) * D, rows_b, D);
    }
    if (live) a.g[G_ACT * a.npad + i] = make_float4(e.s.a0, e.s.a1, e.s.a2, e.s.a3);
  } else if (role == 2) {
    // ================================================================ wave C: gravity + velocity products; the reset sampler
    const bool pool = a.use_pool != 0 && a.auto_reset != 0;
    rl_pool_load(a, G_NX0, il, L.nxt[0], lane, pool);
    rl_pool_load(a, G_NY0, il, L.nxt[1], lane, pool);
    coop_barrier();   // P
    // the sampler job: per lane the episode it samples (NONE: the lane is not part of the job), the Philox words so far, the
    // finished state; `jphase` is wave-uniform: 0 idle, 1..7 the chunk to run next, JOB_DONE finished -> commit
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    constexpr int JOB_DONE = 8;
    float jz[16], ju[2];
#pragma unroll
    for (int k = 0; k < 16; k++) jz[k] = 0.f;
    ju[0] = ju[1] = 0.f;
    uint32_t jx = NONE, jw[20];
#pragma unroll
    for (int k = 0; k < 20; k++) jw[k] = 0u;
    State<float> jns;
    jns.px = jns.py = jns.pz = jns.qw = jns.qx = jns.qy = jns.qz = jns.th1 = jns.th2 = 0.f;
    jns.vx = jns.vy = jns.vz = jns.wx = jns.wy = jns.wz = jns.thd1 = jns.thd2 = jns.a0 = jns.a1 = jns.a2 = jns.a3 = 0.f;
    int jphase = 0;
    for (int t = 0; t < T + (sens ? 1 : 0); t++) {   // (sensor-reading rows: one more half round, the forward dynamics at s_T)
      RL_STAMP(0);
      State<float> s;
      const uint32_t mark = rl_get<false, false>(L.st, lane, s);
      const uint32_t episode = L.info[lane].y;
      // commit a finished job -- in a round in which no lane of the group takes its new state from the pool (the other waves read their
      // slots right now, without looking at tags), and only entries that are still of use: a lane that was reset three times while
      // its job ran (max_steps of a few steps) would otherwise see the entry of an episode long over land in the slot it reads
      if (jphase == JOB_DONE && !__any(mark != 0u ? 1 : 0)) {
        if (jx != NONE && (jx == episode || jx == episode + 1u)) rl_pool_put(L.nxt[jx & 1u], lane, jx, jns);
        jphase = 0;
      }
      const float4 tag0 = L.nxt[0][RL_TAG][lane], tag1 = L.nxt[1][RL_TAG][lane];
      rl_take_reset<false, false>(L, lane, mark, s);
      float s1, c1, s2, c2;
      qsincos(s.th1, &s1, &c1);
      qsincos(s.th2, &s2, &c2);
      const TetherHP<double> th = tether_hp<double>(s1, c1, s2, c2);   // the same pairs the solver wave builds its inverse from
      V3<float> gt, w;
      gravity_body(s, &gt, &w);
      const Inertial<double> in = inertial_wrench_hp<float, double, false>(e.M, s, gt, w, th.d, th.y2);   // the core body's share: wave B
      L.ine[0][lane] = make_double2(in.F.x, in.F.y);
      L.ine[1][lane] = make_double2(in.F.z, in.Tq.x);
      L.ine[2][lane] = make_double2(in.Tq.y, in.Tq.z);
      L.ine[3][lane] = make_double2(in.t1, in.t2);
      RL_STAMP(1);
      coop_barrier();   // 1
      RL_STAMP(2);
      if (pool) {
        if (jphase == 0) {
          // what is missing: the entry of the env's current counter first (it would be sampled inline), else the one after it
          const bool have_c = rl_entry_valid((episode & 1u) ? tag1 : tag0, episode);
          const bool have_n = rl_entry_valid((episode & 1u) ? tag0 : tag1, episode + 1u);
          jx = !have_c ? episode : (!have_n ? episode + 1u : NONE);
          if (__any(jx != NONE ? 1 : 0)) jphase = 1;
        }
        // one Philox block per step (its 32-bit multiplies are quarter rate: ~900 cycles a block), then the Box-Muller pairs, then
        // the transforms: every chunk well inside wave A's phase 2
        switch (jphase) {
          case 1: sample_words<0, 1>(a.seed, (uint32_t)i, jx, jw); jphase = 2; break;
          case 2: sample_words<1, 2>(a.seed, (uint32_t)i, jx, jw); jphase = 3; break;
          case 3: sample_words<2, 3>(a.seed, (uint32_t)i, jx, jw); jphase = 4; break;
          case 4: sample_words<3, 4>(a.seed, (uint32_t)i, jx, jw); jphase = 5; break;
          case 5: sample_words<4, 5>(a.seed, (uint32_t)i, jx, jw); jphase = 6; break;
          case 6: draws_from_words(jw, jz, ju); jphase = 7; break;
          case 7: sample_state<true>(a.sc, jz, ju, jns); jphase = JOB_DONE; break;
          default: break;
        }
      }
      RL_STAMP(3);
      coop_barrier();   // 2
      RL_STAMP(4);
    }
    coop_barrier();   // X
    // hand the pool back to the arena as the per-step kernels expect it: the entry of every env's current counter and of the
    // one after it, complete (what the chunked job had not finished is sampled here, once per fragment); entries are "state only"
    if (pool) {
      const uint32_t episode = L.info[lane].y;
      if (jphase == JOB_DONE && jx != NONE && (jx == episode || jx == episode + 1u)) rl_pool_put(L.nxt[jx & 1u], lane, jx, jns);
#pragma unroll 1
      for (uint32_t d = 0; d < 2; d++) {
        const uint32_t x = episode + d;
        if (!rl_entry_valid(L.nxt[x & 1u][RL_TAG][lane], x)) {
          State<float> ns;
          sample_episode<true>(a, i, x, ns);
          rl_pool_put(L.nxt[x & 1u], lane, x, ns);
        }
      }
      if (live) {
        rl_pool_store(a, G_NX0, i, L.nxt[0], lane);
        rl_pool_store(a, G_NY0, i, L.nxt[1], lane);
      }
    }
  } else {
    // ================================================================ wave D: drag on the tether | rewards, flags, row stores
    float4 act_prev = actions4[il];   // the action of step t - 1 when round t uses it
    coop_barrier();   // P
    const int rows = min(64, n - base_env);
    constexpr int SD = spec_obs_dim<SPEC>() > 0 ? spec_obs_dim<SPEC>() : 4;
    const bool split = !spec_runtime<SPEC>() && rows == 64 && !sens;
    uint32_t mark_before = 0u;   // the lanes the step before the last one reset: their row's sensor entries are due this round
    for (int t = 0; t <= T; t++) {
      RL_STAMP(0);
      State<float> s;
      const uint32_t mark = rl_get<true, true>(L.st, lane, s);
      const uint4 info = L.info[lane];
      State<float> sr = s;   // the state the reward of step t - 1 is of: before a reset
      rl_take_reset<true, false>(L, lane, mark, s);
      // the reading of the step that led here: what the rows of the lanes reset a step earlier still lack (taken here: the solver
      // wave rewrites it in phase 2)
      float4 acc_due = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sens && t >= 1) acc_due = L.accv[lane];
      if (t < T || sens) {
        M3<float> Rd;
        V3<float> vb;
        attitude_min(s, &Rd, &vb);
        const Tether<float> tg = tether_geometry(s.th1, s.th2);
        const Applied<float> ap = applied_tether(e.M, s, mk<float>(s.wx, s.wy, s.wz), vb, tg);
        L.appD[0][lane] = make_float4(ap.F.x, ap.F.y, ap.F.z, ap.t1);
        L.appD[1][lane] = make_float4(ap.Tq.x, ap.Tq.y, ap.Tq.z, ap.t2);
        RL_STAMP(1);
        coop_barrier();   // 1
        RL_STAMP(2);
      }
      // the tile's way out is a round trip through LDS and then the row stores: the reads leave first, the reward runs under them
      float4 tv[SD * 16 / 64 > 0 ? SD * 16 / 64 : 1], tvt = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!FLUSH_B && t >= 2 && split) rl_flush_load<SD>(L.tile[(t - 1) & 1], tv, tvt);
      if (t >= 1) {
        float ref_t[4];
        rc_ref(a, i, (int)info.z - 1, ref0, ref_t);   // the reference the step ran with (episode step before the increment)
        const float act4[4] = {act_prev.x, act_prev.y, act_prev.z, act_prev.w};
        float rw;
        if (spec_term<SPEC>(a) == QD_TERM_SIMPLE) {   // SimpleDrone.step's reward on this model (env_step: 0.1 - |pos - ref|)
          const float dx = sr.px - ref_t[0], dy = sr.py - ref_t[1], dz = sr.pz - ref_t[2];
          rw = 0.1f - qsqrt(dx * dx + dy * dy + dz * dz);
        } else {
          float sv[33];
          M3<float> Rq;
          sr.a0 = sr.a1 = sr.a2 = sr.a3 = 0.f;   // (no reward reads the activations)
          drone_state<float, true>(sr, mk<float>(0.f, 0.f, 0.f), ref_t, e.par, sv, &Rq);
          rw = reward<float>(spec_reward<SPEC>(a), sv, act4, (int)info.z, ref_t, a.max_distance, &Rq);
        }
        if (live) {
          __builtin_nontemporal_store(rw, reward_out + (size_t)(t - 1) * n + i);
          __builtin_nontemporal_store((uint8_t)(info.x & 1u), trunc_out + (size_t)(t - 1) * n + i);
        }
      }
      if (t < T) act_prev = actions4[(size_t)t * n + il];   // for round t + 1: issued ahead of the row stores, in flight across the barrier
      if (!FLUSH_B && t >= 2) {   // row t - 2: wave B built it a round ago
        float* dst = obs + ((size_t)(t - 2) * n + base_env) * D;
        if (split) rl_flush_store<SD>(dst, tv, tvt);
        else {
          if (sens) {   // a lane that step t - 2 reset: its row carries the reading at the new state, i.e. of step t - 1's solve
            if (mark_before != 0u) {
              float* row = L.tile[(t - 1) & 1] + lane * D + acc_at;
              row[0] = acc_due.x; row[1] = acc_due.y; row[2] = acc_due.z;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
          }
          flush_obs_any<SPEC>(L.tile[(t - 1) & 1], dst, rows, D);
        }
      }
      mark_before = mark;
      RL_STAMP(3);
      if (t < T || sens) coop_barrier();   // 2
      RL_STAMP(4);
    }
    coop_barrier();   // X: wave B's last row
    if (!FLUSH_B) {
      if (sens && mark_before != 0u) {   // the last row of a lane the last step reset: the half round's reading (at s_T)
        const float4 x = L.accv[lane];
        float* row = L.tile[T & 1] + lane * D + acc_at;
        row[0] = x.x; row[1] = x.y; row[2] = x.z;
      }
      if (sens) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
      flush_obs_any<SPEC>(L.tile[T & 1], obs + ((size_t)(T - 1) * n + base_env) * D, rows, D);
    }
  }
}

hipError_t launch_rollout_lat(const KArgs& k, int spec, int T, const float* actions, float* obs, float* reward, uint8_t* trunc, hipStream_t stream) {
  KArgs kk = k;
  kk.main_blocks = (k.n + 63) / 64;
  if (kk.main_blocks > 256 || (spec != SPEC_RMA && spec != SPEC_LSTM && spec != SPEC_GENERIC_FS1)) return hipErrorInvalidValue;
  // the workgroup's own sampler (wave C's phase 2): every workgroup has its CU to itself here (launch_rollout_coop on when it pays)
  kk.use_pool = (k.auto_reset && k.sc.random_start != QD_START_FIXED) ? 1 : 0;
  const dim3 grid(kk.main_blocks), block(RL_THREADS);
  (void)hipGetLastError();
  if (spec == SPEC_RMA) hipLaunchKernelGGL((k_rollout_lat<SPEC_RMA>), grid, block, 0, stream, kk, T, actions, obs, reward, trunc);
  else if (spec == SPEC_LSTM) hipLaunchKernelGGL((k_rollout_lat<SPEC_LSTM>), grid, block, 0, stream, kk, T, actions, obs, reward, trunc);
  else hipLaunchKernelGGL((k_rollout_lat<SPEC_GENERIC_FS1>), grid, block, 0, stream, kk, T, actions, obs, reward, trunc);
  return hipGetLastError();
}

#ifdef QD_STAMPS
extern "C" int qd_debug_read_rlstamps(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(qd_rlstamps), sizeof(unsigned long long) * 64 * 4 * 16) == hipSuccess ? 0 : -4;
}
#endif

}  // namespace qd
