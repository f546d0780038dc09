// qd_rollout_coop.hip -- k_rollout_coop: a whole rollout fragment (T env steps) of the training configuration in ONE launch.
//
// Replaces, for T consecutive calls, BaseDroneEnv.vector_step (environments/BaseDroneEnv.py:259-294) as the sampler of
// train_PPO.py / train_RMA.py drives it in 1024-step fragments (train_RMA.py:63): ctrl map, mj_step, counters, truncation,
// reward, (auto-)reset, observation -- the load model with LocalFrameRPYParamsEnv + distance_energy_reward (SPEC_RMA).
//
// Why: one launch per step cannot go below the cost of a dependent launch (1.5 us empty, ~2.0 us with the step's loads and
// stores, profiles/r02_microbench.txt) whatever the kernel does, which caps the 4096-env configuration below 10 % of the HBM
// roofline.  Here a workgroup of FOUR wavefronts (the four SIMDs of one CU) owns 64 envs for all T steps; the state never
// leaves the CU between steps, only actions come in and observation rows / rewards / flags go out (109 B per env-step instead
// of the 309 B a per-step launch moves).  Lane l of every wave works on env l of the group; the waves exchange through LDS at
// two workgroup barriers per step:
//
//   wave A (solver)     mass_factor(s_t), activation filter | reduce_rhs, finish_accel(implicit), integrate, truncation,
//                                                            | in-kernel reset (pool entry from LDS) -> publishes s_{t+1}
//   wave B (applied)    attitude, applied_wrench(s_t)        | -
//   wave C (inertial)   inertial_wrench(s_t)                 | one CHUNK of the workgroup's own reset sampler (below)
//   wave D (epilogue)   one step behind: observation row of s_t (= the row of step t-1) into its LDS tile | reward and flags
//                       of step t-1, 16-byte streaming stores of the 64 rows
//                     barrier 1 ^                                                                  barrier 2 ^
//
// The critical path of a step is {mass_factor | applied_wrench | inertial_wrench} -> solve + integrate; everything else (the
// whole epilogue of k_step_coop's phase 3, the next action's fetch, the sampling of new initial states) runs beside it.
// The arithmetic is k_step_coop's: the same functions of qd_dynamics.h / qd_obsrew.h on the same float32 / float64 values,
// so a fragment equals T x qd_step of the cooperative per-step kernel (tests/test_gpu_parity.py).
//
// The reset sampler.  k_step_coop gets the initial state of an env's next episode from the reset pool in the arena, kept
// filled by sampler workgroups of LATER launches.  Inside one launch there is no later launch, so the pool moves into LDS
// (both slots of every env: the entry of its current episode counter c and of c + 1, loaded from the arena at the start,
// written back at the end) and wave C refills it in the ~1800 cycles per step it would otherwise idle through: a sample
// (5 Philox blocks, 8 Box-Muller pairs, the transforms of sample_state: ~1000 instructions) is cut into three chunks, one
// per step, for all lanes of the group that miss an entry when the job starts; the finished entries are committed at the
// start of a phase 1, when wave A does not read the pool.  A sample is a pure function of (seed, env, episode), so whoever
// computes it and whenever, the result is the same: wave A only ever READS the pool, takes the entry tagged with its counter
// if it is there and samples inline if not (slower, same result) -- the hand-over is the workgroup barrier, there is no
// two-slot global protocol and no fence.  Counted like in the per-step kernels (qd_pool_counters).
#include <cstdlib>

#include "qd_env_device.h"

namespace qd {

constexpr int RC_THREADS = 256;

struct RcLds {
  float4 app[5][64];      // B -> A: Applied (F, t1) (Tq, t2) and the attitude matrix
  double2 ine[4][64];     // C -> A: Inertial (F, Tq, t1, t2)
  float4 st[6][64];       // A -> B, C, D: the state at the start of the next step (after the in-kernel reset of a truncated lane):
                          //   (pos, th1) (quat) (vel, th2) (angvel, thd1) (act) (thd2, -, -, -)
  float4 pre[6][64];      // A -> D: the state after the step and BEFORE the reset, truncated lanes only (the reward is of this state)
  uint4 info[64];         // A -> C, D: (bit 0 truncated | bit 1 reset), episode counter of s_{t+1}, num_steps after the step, -
  float4 nxt[2][5][64];   // the reset pool: slot e & 1 = entry of episode e, planes as in the arena (tag plane last)
  float4 act[64];         // C -> A, D: the controller's action for this step (PID instantiations)
  float4 acc[64];         // A -> D: the accelerometer reading of the previous round (sensor-reading observation variants)
  float4 acc2[64];        // A -> D: the reading at the state the fragment ends in (the extra half round)
  float tile[2][64 * QD_MAX_OBS];   // wave D: the group's observation rows, row-major like the global span; two, because the rows
                                    // of a sensor-reading variant are completed and written out a round after they are built
};

__device__ __forceinline__ void rc_put_state(float4 (*st)[64], int lane, const State<float>& s) {
  st[0][lane] = make_float4(s.px, s.py, s.pz, s.th1);
  st[1][lane] = make_float4(s.qw, s.qx, s.qy, s.qz);
  st[2][lane] = make_float4(s.vx, s.vy, s.vz, s.th2);
  st[3][lane] = make_float4(s.wx, s.wy, s.wz, s.thd1);
  st[4][lane] = make_float4(s.a0, s.a1, s.a2, s.a3);
  st[5][lane] = make_float4(s.thd2, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void rc_get_state(const float4 (*st)[64], int lane, State<float>& s) {
  const float4 p = st[0][lane], q = st[1][lane], v = st[2][lane], w = st[3][lane], c = st[4][lane], x = st[5][lane];
  s.px = p.x; s.py = p.y; s.pz = p.z; s.th1 = p.w;
  s.qw = q.x; s.qx = q.y; s.qy = q.z; s.qz = q.w;
  s.vx = v.x; s.vy = v.y; s.vz = v.z; s.th2 = v.w;
  s.wx = w.x; s.wy = w.y; s.wz = w.z; s.thd1 = w.w;
  s.a0 = c.x; s.a1 = c.y; s.a2 = c.z; s.a3 = c.w;
  s.thd2 = x.x;
}
// a pre-sampled initial state into one slot of the LDS pool, planes and tag as pool_store() writes them to the arena
__device__ __forceinline__ void pool_put_lds(float4 (*slot)[64], int lane, uint32_t episode, const State<float>& s) {
  slot[0][lane] = make_float4(s.px, s.py, s.pz, s.th1);
  slot[1][lane] = make_float4(s.qw, s.qx, s.qy, s.qz);
  slot[2][lane] = make_float4(s.vx, s.vy, s.vz, s.th2);
  slot[3][lane] = make_float4(s.wx, s.wy, s.wz, s.thd1);
  slot[4][lane] = make_float4(s.thd2, __uint_as_float(episode), __uint_as_float(POOL_STATE), 0.f);
}
__device__ __forceinline__ bool rc_entry_valid(float4 tagp, uint32_t episode) {
  return __float_as_uint(tagp.z) != 0u && __float_as_uint(tagp.y) == episode;
}

#ifdef QD_STAMPS
// diagnostic build: cycle stamps of one step in the middle of the fragment, 16 per (workgroup, wave)
__device__ unsigned long long qd_rcstamps[64 * 4 * 16];
#define RC_STAMP(k)                                                                                          \
  do {                                                                                                       \
    if (t == (T >> 1)) {                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      unsigned long long t_;                                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      if (lane == 0 && blockIdx.x < 64) qd_rcstamps[(blockIdx.x * 4 + role) * 16 + (k)] = t_;                \
    }                                                                                                        \
  } while (0)
#else
#define RC_STAMP(k)
#endif

// SPEC_RMA (train_PPO.py / train_RMA.py), SPEC_LSTM (train_LSTM.py) or SPEC_GENERIC_FS1 (any observation / reward of the load
// model, dispatched at run time in wave D), all with skip_steps = 1.
//
// Observation variants that carry the accelerometer (`sens`): the reading in the row of step k is the one mj_step computed in
// that step, at the state the step STARTED from (quirk C-6) -- for the solver wave that is the explicit solve of round k, which
// it evaluates at the start of round k + 1 (its phase 1 has slack) and publishes as L.acc.  The row of a lane that was reset
// in step k instead carries mj_forward's reading at the NEW state with the activations that survived the reset (set_state ->
// mj_forward, mujoco_vecenv.py:396-402): exactly the reading round k + 1 computes anyway.  So the sensor entries of row k are
// final one round later than the rest of it; wave D builds the row into one of two LDS tiles, and patches the three values in
// and writes the tile out a round later.  The fragment's last row needs the reading at s_T: one extra half round (phase 1 and
// the explicit solve, no integration).  No affine sensor form in the pool, no forward dynamics re-run in a truncating lane.
// OCC: workgroups per CU the register budget allows.  Two (<= 256 registers per wave) double the envs in flight at large batches
// and cost SPEC_RMA nothing (251 registers, no scratch); the sensor-carrying and run-time-dispatched instantiations spill a
// dozen registers under that cap, so batches that leave the second slot empty anyway (<= 16384 envs = 256 workgroups) run
// their OCC = 1 instantiation.
//
// PID: the action source is the reference's analytic cascade (qd_pid.h; models/Analytic/*.py wired as attitude_test.py:36-47) instead
// of an action tensor.  The action of step t is a function of s_t, but the step itself does not wait for it: the motors are filters
// (the thrust of step t comes from the activations a_t; u_t only enters a_{t+1}), so wave B evaluates the controller in phase 2 of
// round t, beside wave A's solve (its memory stays in wave B's registers), and the filter is applied a round late -- by wave A to
// the activations it publishes, by wave B to its own copy for the thrust, by wave D for rows that carry the activations -- from the
// action published in L.act before barrier 2.  `actions_out` [T,N,4] (nullable) receives the actions.
template <int SPEC, int OCC, bool PID>
__global__ __launch_bounds__(RC_THREADS, OCC) void k_rollout_coop(KArgs a, int T, const float* __restrict__ actions, float* __restrict__ obs,
                                                             float* __restrict__ reward_out, uint8_t* __restrict__ trunc_out,
                                                             float* __restrict__ actions_out) {
  static_assert(SPEC == SPEC_RMA || SPEC == SPEC_LSTM || SPEC == SPEC_GENERIC_FS1, "persistent fragment kernel: the load model, one substep per step");
  const int D = spec_runtime<SPEC>() ? a.D : spec_obs_dim<SPEC>();
  const bool sens = SPEC == SPEC_RMA ? false : (SPEC == SPEC_LSTM ? true : a.obs_needs_acc != 0);
  const int rounds = T + (sens ? 1 : 0);
  __shared__ RcLds L;
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
    // ================================================================ wave A: factorisation, solve, integration, resets
    float4 act_next = PID ? make_float4(0.f, 0.f, 0.f, 0.f) : actions4[il];
    rc_put_state(L.st, lane, e.s);
    L.info[lane] = make_uint4(0u, e.episode, (uint32_t)e.num_steps, 0u);
    coop_barrier();   // P
    Factor<double> f;
    Rhs<double> r;
    M3<float> R;
    V3<float> w0 = mk<float>(0.f, 0.f, 0.f), acc_last = mk<float>(0.f, 0.f, 0.f);
    bool rst_last = false;
    for (int t = 0; t <= rounds; t++) {
      RC_STAMP(0);
      const bool half = t == T;   // `sens` only: the forward dynamics at s_T for the last row's sensor entries, nothing integrated
      // ---------------------------------------------------------- phase 1
      if (sens && t >= 1) {   // the reading of round t - 1, from that round's factor and right-hand side (before they are replaced)
        // The ONLY place the reading is evaluated -- also for the one at s_T after the extra half round (t == rounds): two inlined
        // copies of the same arithmetic fuse their multiply-adds differently, and a fragment cut would show in the last bit.
        const V3<float> acc_t = rc_sensor(f, r, R, w0);
        if (t == rounds) {
          L.acc2[lane] = make_float4(acc_t.x, acc_t.y, acc_t.z, 0.f);
          if (rst_last) acc_last = acc_t;
        } else {
          acc_last = acc_t;
          L.acc[lane] = make_float4(acc_t.x, acc_t.y, acc_t.z, 0.f);
        }
      }
      // PID: the controller's action of round t - 1 (wave B evaluated it during that round's phase 2, beside this wave's solve).  The
      // motors are filters: u only enters the NEXT state's activations, so nothing of round t - 1 waited for it; it is applied here,
      // before this round publishes them (wave B, which needs a_t for the thrust now, applies the same filter to its own copy).
      if (PID && t >= 1 && t <= T) {
        rc_filter<SPEC>(a, e.M, e.s, L.act[lane]);
      }
      if (t == rounds) break;
      const float4 action = act_next;
      if (!PID && t + 1 < T) act_next = actions4[(size_t)(t + 1) * n + il];   // in flight during this step
      const Tether<float> tg = tether_geometry(e.s.th1, e.s.th2);
      f = mass_factor<true>(e.M, tg, a.h);
      if (!PID && !half) rc_filter<SPEC>(a, e.M, e.s, action);   // ctrl map and activation filter: the part of the Euler step that does not wait for the accelerations
      rc_ref(a, i, e.num_steps, ref0, e.ref);
      // everything the solve reads of the factor exists BEFORE the barrier: the barrier is an asm the compiler moves pure
      // arithmetic across freely, and left alone it sinks two thirds of the factorisation into phase 2 -- onto the critical
      // path, while this wave sits at the barrier waiting for the applied wrench (stamps: phase 2 2600 cycles instead of 1800)
      rc_pin(f.B1); rc_pin(f.B2); rc_pin(f.X1); rc_pin(f.X2); rc_pin(f.rc);
      rc_pin(f.s11, f.s12, f.s22); rc_pin(f.imt, f.m2, f.hb);
      rc_pin(f.Sm); rc_pin(f.kp1); rc_pin(f.kp2);
      rc_pin(f.idet_ex, f.idet_im, f.hb); rc_pin(f.ixx, f.ixy, f.ixz); rc_pin(f.iyy, f.iyz, f.izz);
      asm volatile("" ::"v"(e.s.a0), "v"(e.s.a1), "v"(e.s.a2), "v"(e.s.a3), "v"(e.ref[0]), "v"(e.ref[1]), "v"(e.ref[2]));
      RC_STAMP(1);
      coop_barrier();   // 1
      RC_STAMP(2);
      // ---------------------------------------------------------- phase 2
      {
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
        r = reduce_rhs<true>(f, ap, in);
      }
      w0 = mk<float>(e.s.wx, e.s.wy, e.s.wz);
      if (!half) {
        Accel<float> im;
        V3<double> a0im;
        finish_accel<true, true>(f, r, &a0im, &im.ang, &im.thdd1, &im.thdd2);
        im.lin = mul(R, cvt<float>(a0im));
        integrate_motion<float, true>(e.s, im, a.h);
        e.flags &= ~FLAG_ACC_STALE;
        e.num_steps += 1;
        const int steps_post = e.num_steps;
        bool tr;
        {  // default_termination_fcn / SimpleDrone's rule on the position alone (the tests truncated() / env_step make on the state vector)
          const float dx = e.s.px - e.ref[0], dy = e.s.py - e.ref[1], dz = e.s.pz - e.ref[2];
          const float dist = qsqrt(dx * dx + dy * dy + dz * dz);
          tr = spec_term<SPEC>(a) == QD_TERM_SIMPLE ? dist > 0.5f : (!(dist <= a.max_distance) || e.num_steps >= a.max_steps);
        }
        const bool rst = a.auto_reset && tr;
        if (rst) {
          rc_put_state(L.pre, lane, e.s);
          State<float> ns;   // the new episode's state; the activations carry over (reset_bookkeeping)
          bool taken = false;
          if (a.use_pool) {
            const int sl = (int)(e.episode & 1u);
            const float4 nx4 = L.nxt[sl][4][lane];
            taken = rc_entry_valid(nx4, e.episode);
            if (taken) {
              const float4 p = L.nxt[sl][0][lane], q = L.nxt[sl][1][lane], v = L.nxt[sl][2][lane], w = L.nxt[sl][3][lane];
              ns.px = p.x; ns.py = p.y; ns.pz = p.z; ns.th1 = p.w;
              ns.qw = q.x; ns.qx = q.y; ns.qy = q.z; ns.qz = q.w;
              ns.vx = v.x; ns.vy = v.y; ns.vz = v.z; ns.th2 = v.w;
              ns.wx = w.x; ns.wy = w.y; ns.wz = w.z; ns.thd1 = w.w;
              ns.thd2 = nx4.x;
            }
          }
          if (!taken) sample_episode<true>(a, i, e.episode, ns);
          ns.a0 = e.s.a0; ns.a1 = e.s.a1; ns.a2 = e.s.a2; ns.a3 = e.s.a3;
          e.s = ns;
          reset_bookkeeping(e.s, e.episode, e.num_steps);
          // without the sensor in the row the stored reading is only marked stale (the next step, or a getter that runs first,
          // recomputes it); with it, the next round's reading takes its place (below)
          if (!sens) e.flags |= FLAG_ACC_STALE;
          if (a.use_pool && live) pool_count(a, taken);
        }
        rst_last = rst;
        rc_put_state(L.st, lane, e.s);
        L.info[lane] = make_uint4((tr ? 1u : 0u) | (rst ? 2u : 0u), e.episode, (uint32_t)steps_post, 0u);
      }
      // (the half round stops at the right-hand side: the reading at s_T -- the sensor entries of the last row where the last
      // step reset the lane, and what mj_forward leaves in the arena -- is evaluated at the top of the loop like every other)
      RC_STAMP(3);
      coop_barrier();   // 2
      RC_STAMP(4);
    }
    if (sens) coop_barrier();   // E: L.acc2 is published
    // the fragment's last step leaves what a per-step launch leaves: the state, and the accelerometer reading of that step
    // (quirk C-6: the reading of the state the step STARTED from; where a reset replaced or invalidated it, see above)
    if (live) {
      e.acc = sens ? acc_last : rc_sensor(f, r, R, w0);
      store_env(a, i, e);
    }
  } else if (role == 1) {
    // ================================================================ wave B: thrust + drag on the three bodies; the PID cascade
    // PID instantiations: the controller pair of this lane's env lives in this wave's registers (arena planes C0..C3 between launches).
    // It is evaluated on s_t in PHASE 2 of round t, which this wave otherwise idles through: the action u_t is not needed before the
    // activations of s_{t+1} are (see wave A), i.e. by this wave's own thrust at the start of round t + 1.  (Round 3 first had the
    // cascade in wave C's phase 1, in front of barrier 1: 1.92 us per step against 1.35 with given actions.)
    PidState<float> pc;
    pid_reset(pc);
    if (PID) load_pid(a, il, pc);
    float4 u_last = make_float4(0.f, 0.f, 0.f, 0.f);
    coop_barrier();   // P
    for (int t = 0; t < rounds; t++) {
      RC_STAMP(0);
      State<float> s;
      rc_get_state(L.st, lane, s);
      if (PID && t >= 1) {   // the published activations are one filter step behind: a_t = filter(a_{t-1}, u_{t-1})
        rc_filter<SPEC>(a, e.M, s, u_last);
      }
      const Tether<float> tg = tether_geometry(s.th1, s.th2);
      const Att<float> at = attitude(s);
      const Applied<float> ap = applied_wrench(e.M, s, at, tg);
      L.app[0][lane] = make_float4(ap.F.x, ap.F.y, ap.F.z, ap.t1);
      L.app[1][lane] = make_float4(ap.Tq.x, ap.Tq.y, ap.Tq.z, ap.t2);
      L.app[2][lane] = make_float4(at.R.m00, at.R.m01, at.R.m02, at.R.m10);
      L.app[3][lane] = make_float4(at.R.m11, at.R.m12, at.R.m20, at.R.m21);
      L.app[4][lane] = make_float4(at.R.m22, 0.f, 0.f, 0.f);
      uint4 info = make_uint4(0u, 0u, 0u, 0u);
      if (PID) info = L.info[lane];   // of s_t (wave A rewrites it in phase 2)
      RC_STAMP(1);
      coop_barrier();   // 1
      RC_STAMP(2);
      if (PID && t < T) {
        // A lane reset in the last step starts with fresh controller objects; the waypoint is that of the episode step s_t is at.
        const bool rst = (info.x & 2u) != 0u;
        if (rst) pid_reset(pc);
        EnvRegs ed;
        ed.s = s;
        rc_ref(a, i, rst ? 0 : (int)info.z, ref0, ed.ref);
#pragma unroll
        for (int k = 0; k < 6; k++) ed.par[k] = e.par[k];
        u_last = pid_env_action(pc, ed);
        L.act[lane] = u_last;
        if (actions_out && live) reinterpret_cast<float4*>(actions_out)[(size_t)t * n + i] = u_last;
      }
      RC_STAMP(3);
      coop_barrier();   // 2
      RC_STAMP(4);
    }
    if (sens) coop_barrier();   // E
    if (PID && live) {   // the controller memory back to the arena (fresh objects where the last step reset the lane)
      if ((L.info[lane].x & 2u) != 0u) pid_reset(pc);
      store_pid(a, i, pc);
    }
  } else if (role == 2) {
    // ================================================================ wave C: gravity + velocity products; the reset sampler
    const bool pool = a.use_pool != 0 && a.auto_reset != 0;
    {
#pragma unroll
      for (int k = 0; k < 5; k++) {
        L.nxt[0][k][lane] = pool ? a.g[(G_NX0 + k) * a.npad + il] : make_float4(0.f, 0.f, 0.f, 0.f);
        L.nxt[1][k][lane] = pool ? a.g[(G_NY0 + k) * a.npad + il] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
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
    for (int t = 0; t < rounds; t++) {
      RC_STAMP(0);
      if (jphase == JOB_DONE) {   // commit: wave A is in its phase 1 and does not read the pool
        if (jx != NONE) pool_put_lds(L.nxt[jx & 1u], lane, jx, jns);
        jphase = 0;
      }
      State<float> s;
      rc_get_state(L.st, lane, s);
      const uint4 info = L.info[lane];
      const uint32_t episode = info.y;
      const float4 tag0 = L.nxt[0][4][lane], tag1 = L.nxt[1][4][lane];
      const Tether<float> tg = tether_geometry(s.th1, s.th2);
      V3<float> gt, w;
      gravity_body(s, &gt, &w);
      const Inertial<double> in = inertial_wrench(e.M, s, gt, w, tg);
      L.ine[0][lane] = make_double2(in.F.x, in.F.y);
      L.ine[1][lane] = make_double2(in.F.z, in.Tq.x);
      L.ine[2][lane] = make_double2(in.Tq.y, in.Tq.z);
      L.ine[3][lane] = make_double2(in.t1, in.t2);
      RC_STAMP(1);
      coop_barrier();   // 1
      RC_STAMP(2);
      if (pool) {
        if (jphase == 0) {
          // what is missing: the entry of the env's current counter first (it would be sampled inline), else the one after it
          const bool have_c = rc_entry_valid((episode & 1u) ? tag1 : tag0, episode);
          const bool have_n = rc_entry_valid((episode & 1u) ? tag0 : tag1, episode + 1u);
          jx = !have_c ? episode : (!have_n ? episode + 1u : NONE);
          if (__any(jx != NONE ? 1 : 0)) jphase = 1;
        }
        // one Philox block per step (its 32-bit multiplies are quarter rate: ~900 cycles a block), then the Box-Muller pairs,
        // then the transforms: every chunk well inside wave A's phase 2 (three blocks at once made this wave the last one at
        // barrier 2 in half of the steps)
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
      RC_STAMP(3);
      coop_barrier();   // 2
      RC_STAMP(4);
    }
    if (sens) coop_barrier();   // E
    // hand the pool back to the arena as the per-step kernels expect it: the entry of every env's current counter and of the
    // one after it, complete (what the chunked job had not finished is sampled here, once per fragment); entries are "state
    // only" (POOL_STATE): a per-step kernel of a sensor-reading configuration adds the second stage itself
    if (pool) {
      if (jphase == JOB_DONE && jx != NONE) pool_put_lds(L.nxt[jx & 1u], lane, jx, jns);
      const uint32_t episode = L.info[lane].y;
#pragma unroll 1
      for (uint32_t d = 0; d < 2; d++) {
        const uint32_t x = episode + d;
        if (!rc_entry_valid(L.nxt[x & 1u][4][lane], x)) {
          State<float> ns;
          sample_episode<true>(a, i, x, ns);
          pool_put_lds(L.nxt[x & 1u], lane, x, ns);
        }
      }
      if (live) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
          a.g[(G_NX0 + k) * a.npad + i] = L.nxt[0][k][lane];
          a.g[(G_NY0 + k) * a.npad + i] = L.nxt[1][k][lane];
        }
      }
    }
  } else {
    // ================================================================ wave D: observation rows, rewards, flags -- one round behind
    // (the sensor entries of a row and its write-out: two rounds behind, see the head of the kernel)
    float4 act_prev = PID ? make_float4(0.f, 0.f, 0.f, 0.f) : actions4[il];   // the action of step t - 1 when iteration t uses it
    coop_barrier();   // P
    const int acc_at = sens ? rc_acc_slot(spec_obs<SPEC>(a)) : -1;
    const int rows = min(64, n - base_env);
    V3<float> acc_prev = mk<float>(0.f, 0.f, 0.f);
    bool rst_prev = false;
    for (int t = 0; t <= rounds; t++) {
      RC_STAMP(0);
      float sv[33];
      M3<float> Rq;
      float ref_t[4];
      uint4 info = make_uint4(0u, 0u, 0u, 0u);
      bool rst = false;
      float rw = 0.f;
      const bool row_now = t >= 1 && t <= T;   // the row of step t - 1 is the observation of s_t
      float* tile_now = L.tile[(t - 1) & 1];
      // PID: the action of step t - 1 (wave B published it in that round's phase 2)
      if (PID && row_now) act_prev = L.act[lane];
      if (row_now) {
        EnvRegs ed;   // what write_obs_row reads of an env: its state and reference
        rc_get_state(L.st, lane, ed.s);
        if (PID) {   // the published activations are one filter step behind (wave A applies u_{t-1} at the top of this round)
          rc_filter<SPEC>(a, e.M, ed.s, act_prev);
        }
        info = L.info[lane];
        rst = (info.x & 2u) != 0u;
        rc_ref(a, i, (int)info.z - 1, ref0, ref_t);   // the reference the step ran with (episode step before the increment)
        ed.ref[0] = ref_t[0]; ed.ref[1] = ref_t[1]; ed.ref[2] = ref_t[2]; ed.ref[3] = ref_t[3];
        if (rst && a.ref_mode != QD_REF_STATIC) moving_reference(a, i, 0, ed.ref);   // a new episode's first row
        drone_state<float, true>(ed.s, mk<float>(0.f, 0.f, 0.f), ed.ref, e.par, sv, &Rq);
        write_obs_row<true, SPEC>(a, ed, sv, &Rq, tile_now + lane * D);
        // The reward of step t - 1, here and not behind barrier 1: for a truncated lane it is of the state BEFORE its reset, L.pre,
        // which wave A rewrites in phase 2 of THIS round for the lanes that truncate in step t -- a lane that resets in two
        // consecutive steps (max_steps <= 1, a start sampled outside max_distance) would otherwise race with its own next reset.
        const float act4[4] = {act_prev.x, act_prev.y, act_prev.z, act_prev.w};
        const bool simple = spec_term<SPEC>(a) == QD_TERM_SIMPLE;
        if (simple) {   // SimpleDrone.step's reward on this model (env_step: 0.1 - |pos - ref|)
          const float dx = sv[0] - ref_t[0], dy = sv[1] - ref_t[1], dz = sv[2] - ref_t[2];
          rw = 0.1f - qsqrt(dx * dx + dy * dy + dz * dz);
        } else {
          rw = reward<float>(spec_reward<SPEC>(a), sv, act4, (int)info.z, ref_t, a.max_distance, &Rq);
        }
        if (__any(rst ? 1 : 0)) {
          State<float> p;
          rc_get_state(L.pre, lane, p);
          float sv2[33];
          M3<float> Rq2;
          drone_state<float, true>(p, mk<float>(0.f, 0.f, 0.f), ref_t, e.par, sv2, &Rq2);
          float rw2;
          if (simple) {
            const float dx = sv2[0] - ref_t[0], dy = sv2[1] - ref_t[1], dz = sv2[2] - ref_t[2];
            rw2 = 0.1f - qsqrt(dx * dx + dy * dy + dz * dz);
          } else {
            rw2 = reward<float>(spec_reward<SPEC>(a), sv2, act4, (int)info.z, ref_t, a.max_distance, &Rq2);
          }
          if (rst) rw = rw2;
        }
      }
      RC_STAMP(1);
      if (t < rounds) coop_barrier();   // 1
      else if (sens) coop_barrier();    // E: the reading at s_T is published
      RC_STAMP(2);
      V3<float> acc_new = mk<float>(0.f, 0.f, 0.f);
      if (sens && t >= 1) {   // the reading of round t - 1 (wave A's phase 1 of this round); in the last iteration the one at s_T
        const float4 x = (t == rounds) ? L.acc2[lane] : L.acc[lane];
        acc_new = mk<float>(x.x, x.y, x.z);
      }
      if (row_now) {
        if (live) {
          __builtin_nontemporal_store(rw, reward_out + (size_t)(t - 1) * n + i);
          __builtin_nontemporal_store((uint8_t)(info.x & 1u), trunc_out + (size_t)(t - 1) * n + i);
        }
        if (!sens) {
          // the tile was written by this wave's own lanes in phase 1; in the last iteration no workgroup barrier lies between
          if (t == rounds) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
          flush_obs_any<SPEC>(tile_now, obs + ((size_t)(t - 1) * n + base_env) * D, rows, D);
        }
      }
      if (sens && t >= 2) {   // row t - 2: its sensor entries are final now
        float* tile_then = L.tile[t & 1];
        const V3<float> af = rst_prev ? acc_new : acc_prev;
        float* row = tile_then + lane * D + acc_at;
        row[0] = af.x; row[1] = af.y; row[2] = af.z;
        __builtin_amdgcn_wave_barrier();
        flush_obs_any<SPEC>(tile_then, obs + ((size_t)(t - 2) * n + base_env) * D, rows, D);
      }
      acc_prev = acc_new;
      rst_prev = rst;
      if (!PID && t < T) act_prev = actions4[(size_t)t * n + il];   // for iteration t + 1: in flight across the barrier
      RC_STAMP(3);
      if (t < rounds) coop_barrier();   // 2
      RC_STAMP(4);
    }
  }
}

// ---- SimpleDrone (BASELINE config 1 / 2): two wavefronts per 64 envs -----------------------------------------------------------
// The single-body step is ~440 vector instructions of physics (two substeps at 1 kHz, SimpleDrone.py:54-61) and ~300 of
// epilogue (the 6-value observation: a quaternion to matrix and three inverse trigonometric functions, SimpleDrone.py:94-98; the
// row's way out through LDS).  k_rollout runs both in one wavefront; here a PHYSICS wave integrates, decides termination and the
// reward (both functions of the distance it has anyway) and publishes the state, and an EPILOGUE wave, one step behind, turns
// the published state into the row and streams it out.  One barrier per step, the published state double-buffered.
struct RpLds {
  float4 st[2][2][64];   // [buffer][(pos, -) / (quat)][lane]: what simple_obs reads of an env
  float tile[64 * 8];
};

__global__ __launch_bounds__(128) void k_rollout_pair(KArgs a, int T, const float* __restrict__ actions, float* __restrict__ obs,
                                                      float* __restrict__ reward_out, uint8_t* __restrict__ trunc_out) {
  __shared__ RpLds L;
  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int base_env = blockIdx.x * 64;
  const int i = base_env + lane;
  const bool live = i < a.n;
  const int il = live ? i : a.n - 1;
  const int n = a.n;
  if (role == 0) {
    // ================================================================ physics
    EnvRegs e;
    load_env<false, false, false>(a, il, e);
    const float4* actions4 = reinterpret_cast<const float4*>(actions);
    float4 act_next = actions4[il];
    for (int t = 0; t < T; t++) {
      const float4 action = act_next;
      if (t + 1 < T) act_next = actions4[(size_t)(t + 1) * n + il];
      if (a.ref_mode != QD_REF_STATIC) moving_reference(a, i, e.num_steps, e.ref);
      const float c0 = qclamp(action.x, 0.f, 1.f), c1 = qclamp(action.y, 0.f, 1.f), c2 = qclamp(action.z, 0.f, 1.f), c3 = qclamp(action.w, 0.f, 1.f);
      e.acc = substep<float, false>(e.M, e.s, c0, c1, c2, c3, a.h);
      e.acc = substep<float, false>(e.M, e.s, c0, c1, c2, c3, a.h);
      e.flags &= ~FLAG_ACC_STALE;
      e.num_steps += 1;
      // SimpleDrone.step: terminated = |pos - ref| > 0.5, reward = 0.1 - |pos - ref| (SimpleDrone.py:57-60)
      const float dx = e.s.px - e.ref[0], dy = e.s.py - e.ref[1], dz = e.s.pz - e.ref[2];
      const float d = qsqrt(dx * dx + dy * dy + dz * dz);
      const bool tr = d > 0.5f;
      if (live) {
        __builtin_nontemporal_store(0.1f - d, reward_out + (size_t)t * n + i);
        __builtin_nontemporal_store((uint8_t)(tr ? 1 : 0), trunc_out + (size_t)t * n + i);
      }
      if (a.auto_reset && tr) {
        if (live) reset_in_step<false, true>(a, i, e);
        if (a.ref_mode != QD_REF_STATIC) moving_reference(a, i, 0, e.ref);
      }
      const int b = t & 1;
      L.st[b][0][lane] = make_float4(e.s.px, e.s.py, e.s.pz, 0.f);
      L.st[b][1][lane] = make_float4(e.s.qw, e.s.qx, e.s.qy, e.s.qz);
      coop_barrier();
    }
    if (live) store_env(a, i, e);
  } else {
    // ================================================================ epilogue, one step behind
    for (int t = 0; t <= T; t++) {
      if (t >= 1) {
        const int b = (t - 1) & 1;
        const float4 p = L.st[b][0][lane], q = L.st[b][1][lane];
        State<float> s;
        s.px = p.x; s.py = p.y; s.pz = p.z; s.qw = q.x; s.qx = q.y; s.qy = q.z; s.qz = q.w;
        float o[6];
        simple_obs<float>(s, o);
        float* row = L.tile + lane * 6;
#pragma unroll
        for (int k = 0; k < 6; k++) row[k] = o[k];
        __builtin_amdgcn_wave_barrier();
        flush_obs_any<SPEC_SIMPLE>(L.tile, obs + ((size_t)(t - 1) * n + base_env) * 6, min(64, n - base_env), 6);
        __builtin_amdgcn_wave_barrier();
      }
      if (t < T) coop_barrier();
    }
  }
}

hipError_t launch_rollout_pair(const KArgs& k, int T, const float* actions, float* obs, float* reward, uint8_t* trunc, hipStream_t stream) {
  KArgs kk = k;
  kk.main_blocks = (k.n + 63) / 64;
  (void)hipGetLastError();
  hipLaunchKernelGGL(k_rollout_pair, dim3(kk.main_blocks), dim3(128), 0, stream, kk, T, actions, obs, reward, trunc);
  return hipGetLastError();
}

hipError_t launch_rollout_coop(const KArgs& k, int spec, int T, const float* actions, float* obs, float* reward, uint8_t* trunc, hipStream_t stream,
                               bool pid, float* actions_out) {
  KArgs kk = k;
  kk.main_blocks = (k.n + 63) / 64;
  // The workgroup's own sampler (wave C's phase 2) pays while a workgroup has its CU to itself (<= 256 workgroups = 16384 envs):
  // there a truncating lane that samples inline stalls its whole workgroup for ~5500 cycles.  With two workgroups per CU the other
  // workgroup fills that hole, and a sampler chunk in EVERY step costs more issue slots than it saves: measured in the steady
  // state of config 3 (0.45 truncations per step per 64 envs, tests/diag_persistent_big.py with QD_DIAG_WARM_STEPS=1536) 2.95 us
  // per step with the sampler against 2.62 without at 32768 envs, 91.7 against 72.8 at 2^20.  Entries are a pure function of
  // (seed, env, episode): results are the same either way, and the per-step kernels can use or ignore what is left in the arena.
  kk.use_pool = (k.auto_reset && k.sc.random_start != QD_START_FIXED && k.n <= 256 * 64) ? 1 : 0;
  const dim3 grid(kk.main_blocks), block(RC_THREADS);
  (void)hipGetLastError();
  const bool two = kk.main_blocks > 256;   // more workgroups than CUs: the second slot per CU is worth its register cap
#define RC_LAUNCH(SPECV, OCCV, PIDV) \
  hipLaunchKernelGGL((k_rollout_coop<SPECV, OCCV, PIDV>), grid, block, 0, stream, kk, T, actions, obs, reward, trunc, actions_out)
#define RC_BY_OCC(SPECV, PIDV) do { if (two) RC_LAUNCH(SPECV, 2, PIDV); else RC_LAUNCH(SPECV, 1, PIDV); } while (0)
  if (!pid) {
    if (spec == SPEC_RMA) RC_LAUNCH(SPEC_RMA, 2, false);
    else if (spec == SPEC_LSTM) RC_BY_OCC(SPEC_LSTM, false);
    // (the run-time-dispatched instantiation spills 65 registers under the two-workgroups-per-CU cap and is still the faster one
    // where the second workgroup finds room: 8.7 against 12.0 us per step at 65536 envs, 145 against 205 at 2^20 with the
    // spill-free OCC = 1 build, LocalFrameRmParamsEnv + reward_3, tests/diag_generic_big.py, profiles/r04_notes.md)
    else if (spec == SPEC_GENERIC_FS1) RC_BY_OCC(SPEC_GENERIC_FS1, false);
    else return hipErrorInvalidValue;
  } else {
    if (spec == SPEC_RMA) RC_BY_OCC(SPEC_RMA, true);
    else if (spec == SPEC_LSTM) RC_BY_OCC(SPEC_LSTM, true);
    else if (spec == SPEC_GENERIC_FS1) RC_BY_OCC(SPEC_GENERIC_FS1, true);
    else return hipErrorInvalidValue;
  }
#undef RC_BY_OCC
#undef RC_LAUNCH
  return hipGetLastError();
}

#ifdef QD_STAMPS
extern "C" int qd_debug_read_rcstamps(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(qd_rcstamps), sizeof(unsigned long long) * 64 * 4 * 16) == hipSuccess ? 0 : -4;
}
#endif

}  // namespace qd
