// qd_rollout_fused.hip -- the closed policy -> env loop (rollout.py:64-85 / the sampler's fragments) with the env step
// running BESIDE the forward pass instead of after it.
//
// What makes that possible is in the model, not in the kernel: the motors are first-order filters (dyntype filter, env_gen.py), so
// the force of step t comes from the activations a_t, and the action u_t only enters the activations of the NEXT state,
// a_{t+1} = a_t + h (clamp(u_t) - a_t) / tau.  Position, attitude, velocities and hinge angles of s_{t+1} -- everything an
// observation row of the sensor-free variants reads -- do not depend on u_t at all, and neither does the truncation test; only
// the reward (its energy term) and the four activations do.  So while waves 0..3 of a workgroup run the network on observation
// o_t (k_policy_static's code: float16-pair MFMA, activations in LDS), waves 4..7 advance the same 16 envs from s_t to s_{t+1} in the
// roles of k_rollout_coop (A factorisation / solve / integration / resets, B applied wrench, C inertial wrench, D observation
// row), 16 lanes each, and have o_{t+1} in LDS long before the network has u_t.  What does read the action runs a pass LATE, off
// the critical path: wave A applies the filter between the next pass's two gather barriers (waves B and C read the activations
// behind the second), wave D evaluates reward and flags behind them.  A step costs the forward pass, not forward pass + env step
// (round 2's k_rollout_fused, one wave of 16 lanes after the network: 11.8 us per step at 4096 envs, 3.6 of them the env phase;
// this kernel: 7.9).
//
// gfx950 has ONE barrier per workgroup, so the env waves pass exactly the barriers the network executes: two at the input
// gather, one per executed layer, one behind the outputs.  The env stages sit in front of the first three layer barriers: stage 1
// (the three role waves) before the first, stage 2 (wave A: solve, integration, termination, reset) before the second -- started
// as soon as the other role waves have counted themselves in, with the first layer barrier passed in its middle --, wave D's
// staging of the next pass's inputs before the third; wave D's row waits for no barrier but polls a tag wave A writes behind the
// new state.  For RMA_full the layers behind those barriers take 1.4 / 2.9 / 1.3 us, the stages 0.5 / 2.0 / 1.1 + 0.4 (stamps,
// profiles/r03_policy_loop_timeline.txt).  The env waves run at raised priority (s_setprio 3).
// The 23-value rows of train_LSTM.py's configuration carry the activations and the accelerometer: see SPEC_LSTM in the kernel.
// Arithmetic: qd_dynamics.h's latency arrangement (k_rollout_lat's: mass_inverse / solve_inv5, the accelerometer's explicit
// accelerations by explicit_from_implicit; a reset lane's affine sensor form stays sensor_affine) -- equal to the per-step kernels to
// rounding (tests/test_gpu_policy.py compares with the two-launch loop).  Resets sample inline (a pure function of seed, env,
// episode: the same states the pool would serve), inside the longest layer's window.
#include <cstdlib>

#define QD_POL_SECOND_UNIT
// weights of the next layer requested ahead of the barrier: two k-blocks here (four in k_policy_static) -- under this kernel's
// 256-register cap four cost 70 spilled registers and 1.1 us per step (6.84 against 5.74, RMA_full at 4096 envs)
#ifndef QD_POL_SPF
#define QD_POL_SPF 2
#endif
#include "qd_env_device.h"
#include "qd_policy_static.h"

namespace qd {

constexpr int FP_THREADS = 2 * POL_THREADS;   // waves 0..3 the network, waves 4..7 the env roles

struct FpLds {                 // the env waves' hand-over: one column per env of the tile
  float4 app[5][POL_TILE];     // B -> A: Applied (F, t1) (Tq, t2), the attitude matrix
  double2 ine[4][POL_TILE];    // C -> A: Inertial (F, Tq, t1, t2)
  float4 st[6][POL_TILE];      // A -> B, C, D: the state the next step starts from (rc_put_state's planes)
  float4 pre[6][POL_TILE];     // A -> D: the state before the reset of a truncated lane (its reward is of this state)
  uint4 info[POL_TILE];        // A -> D: (bit 0 truncated | bit 1 reset), episode counter, num_steps after the step, -
  float4 acc[POL_TILE];        // A -> D: the accelerometer reading in the row of this step (sensor-carrying rows; not for reset lanes)
  int tag;                     // A -> D: t + 1 once st / pre / info of step t are published (wave D polls it: no barrier of its own to wait at)
  unsigned int ready;          // B, C, D -> A: + 1 each once a pass's wrenches are written / its reward is settled (3 (t + 1): stage 2 of pass t may start)
};

__device__ __forceinline__ void fp_put_state(float4 (*st)[POL_TILE], int lane, const State<float>& s) {
  st[0][lane] = make_float4(s.px, s.py, s.pz, s.th1);
  st[1][lane] = make_float4(s.qw, s.qx, s.qy, s.qz);
  st[2][lane] = make_float4(s.vx, s.vy, s.vz, s.th2);
  st[3][lane] = make_float4(s.wx, s.wy, s.wz, s.thd1);
  st[4][lane] = make_float4(s.a0, s.a1, s.a2, s.a3);
  st[5][lane] = make_float4(s.thd2, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void fp_get_state(const float4 (*st)[POL_TILE], int lane, State<float>& s) {
  const float4 p = st[0][lane], q = st[1][lane], v = st[2][lane], w = st[3][lane], c = st[4][lane], x = st[5][lane];
  s.px = p.x; s.py = p.y; s.pz = p.z; s.th1 = p.w;
  s.qw = q.x; s.qx = q.y; s.qy = q.z; s.qz = q.w;
  s.vx = v.x; s.vy = v.y; s.vz = v.z; s.th2 = v.w;
  s.wx = w.x; s.wy = w.y; s.wz = w.z; s.thd1 = w.w;
  s.a0 = c.x; s.a1 = c.y; s.a2 = c.z; s.a3 = c.w;
  s.thd2 = x.x;
}

// the barriers s_run<A, LC, J0> executes in one pass: one per op that runs and is not folded into the layer before it
template <class A, int I>
__device__ __forceinline__ int fp_run_barriers_from(bool want_value, unsigned skip_ops) {
  if constexpr (I < A::prog.n_ops) {
    constexpr SOp op = A::prog.op[I];
    constexpr bool folded = op.kind != POL_DENSE && I > 0 && sp_fused_affine(A::prog, I > 0 ? I - 1 : 0);
    const bool runs = (want_value || !(op.flags & POL_FLAG_VALUE_ONLY)) && !((skip_ops >> I) & 1u);
    return ((runs && !folded) ? 1 : 0) + fp_run_barriers_from<A, I + 1>(want_value, skip_ops);
  } else {
    return 0;
  }
}
template <class A>
__device__ __forceinline__ int fp_run_barriers(bool want_value, unsigned skip_ops) {
  return fp_run_barriers_from<A, sp_leading_copies(A::prog)>(want_value, skip_ops);
}
template <class A>
constexpr int fp_min_barriers() {   // actor only, parameter encoder skipped: the fewest a step can have
  int nb = 0;
  for (int I = sp_leading_copies(A::prog); I < A::prog.n_ops; I++) {
    const bool runs = !(A::prog.op[I].flags & POL_FLAG_VALUE_ONLY) && !((fused_const_ops<A> >> I) & 1u);
    const bool folded = A::prog.op[I].kind != POL_DENSE && I > 0 && sp_fused_affine(A::prog, I - 1);
    if (runs && !folded) nb++;
  }
  return nb;
}

// The next pass's gathered inputs, prepared by wave D while the network is still busy: buffers 0 and 1 as the leading COPY_OBS ops
// fill them (zero elsewhere), in a staging copy the network moves in with one float4 per thread -- the gather itself (index
// arithmetic with divisions by the slice widths, two rounds of LDS latency) took 1.0 of the pass's 8.6 us at its head.
template <class A, int I, int LC>
__device__ __forceinline__ void fp_stage_obs(const float* otile, float* stage, int rows, int lane) {
  if constexpr (I < LC) {
    constexpr SOp op = A::prog.op[I];
    static_assert(op.kind == POL_COPY_OBS || op.kind == POL_COPY_PREV, "feed-forward networks: the leading ops gather from the row and the previous action");
    if constexpr (op.kind == POL_COPY_OBS) {
      constexpr int n = op.in_dim, ld = sp_ld(A::prog, op.out_buf), out_base = sp_base(A::prog, op.out_buf) + op.out_off, od = A::prog.obs_dim;
      for (int k = lane; k < POL_TILE * n; k += 64) {
        const int r = k / n, c = k - r * n;
        stage[out_base + r * ld + c] = r < rows ? otile[r * od + op.in_off + c] : 0.f;
      }
    }
    fp_stage_obs<A, I + 1, LC>(otile, stage, rows, lane);
  }
}
// the leading COPY_OBS destinations of NV consecutive row entries (what a row carries of the action: known a pass late)
template <class A, int I, int LC, int NV>
__device__ __forceinline__ void fp_patch_obs(float* lds, int r, int col0, const float (&w)[NV]) {
  if constexpr (I < LC) {
    constexpr SOp op = A::prog.op[I];
    if constexpr (op.kind == POL_COPY_OBS) {
      constexpr int ld = sp_ld(A::prog, op.out_buf), out_base = sp_base(A::prog, op.out_buf) + op.out_off;
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const int c = col0 + j;
        if (c >= op.in_off && c < op.in_off + op.in_dim) lds[out_base + r * ld + (c - op.in_off)] = w[j];
      }
    }
    fp_patch_obs<A, I + 1, LC, NV>(lds, r, col0, w);
  }
}

template <class A>
constexpr int fp_prev_op() {   // the leading COPY_PREV op (-1: none, -2: more than one)
  int at = -1;
  for (int I = 0; I < sp_leading_copies(A::prog); I++)
    if (A::prog.op[I].kind == POL_COPY_PREV) at = at == -1 ? I : -2;
  return at;
}
#ifdef QD_STAMPS
// diagnostic build: s_memrealtime (10 ns) stamps of the last step, [0..15] network (thread 0), [16 + 8 role + k] the env waves
__device__ unsigned long long qd_fpstamps[64];
#define FP_STAMP(k)                                                                                     \
  do {                                                                                                  \
    if (blockIdx.x == 0 && lane == 0 && (k) < 64) qd_fpstamps[(k)] = __builtin_amdgcn_s_memrealtime();  \
  } while (0)
#else
#define FP_STAMP(k)
#endif

template <int SPEC, class A>
__global__ __launch_bounds__(FP_THREADS) void k_rollout_fused_pipe(KArgs a, PolArgs p, int T, PolSample smp, const float* __restrict__ obs0,
                                                                   const float* __restrict__ prev0, float* __restrict__ obs,
                                                                   float* __restrict__ actions, float* __restrict__ reward,
                                                                   uint8_t* __restrict__ trunc, float* __restrict__ logp,
                                                                   float* __restrict__ logits, float* __restrict__ value) {
  static_assert(SPEC == SPEC_RMA || SPEC == SPEC_LSTM, "the observation row must not read the activations");
  // SPEC_LSTM rows (LocalFrameFullStateEnv, 23 values) carry two things more.  The accelerometer: the reading mj_step computed in
  // the step, at the state it STARTED from (quirk C-6) -- the explicit solve of the same pass, no action in it; only a lane that was
  // reset carries mj_forward's reading at the NEW state with the activations that survived the reset, i.e. the filtered ones, and
  // that reading is affine in them (sensor_affine: wave A prepares the constant and the four columns with the new state).  And the
  // four activations themselves.  Both are known when wave A applies the filter, a pass late: it writes them into the network's
  // input buffers (the gather has run, the first layer has not) and into the entries of the global row that wave D left out.
  constexpr bool sens = SPEC == SPEC_LSTM;
  constexpr int acc_at = 12, act_at = 15;   // rc_acc_slot(OBS_FULLSTATE); the activations follow the reading (qd_obsrew.h)
  static_assert(fp_min_barriers<A>() >= 3, "three layer barriers per step carry the three env stages");
  static_assert(sp_hw(A::prog, 0) == 0 && sp_hw(A::prog, 1) == 0, "buffers 0 and 1 are written outside the program (staged inputs, patches): float32 only");
  extern __shared__ float lds[];
  __shared__ FpLds L;
  constexpr int IN_FLOATS_ = POL_TILE * (sp_ld(A::prog, 0) + sp_ld(A::prog, 1));
  static_assert(sp_base(A::prog, 0) == 0 && sp_base(A::prog, 1) == POL_TILE * sp_ld(A::prog, 0) && IN_FLOATS_ % 4 == 0, "buffers 0 and 1 open the activation area");
  __shared__ __attribute__((aligned(16))) float stage[IN_FLOATS_];   // wave D -> network: the gathered inputs of the next pass
  constexpr int ACT = sp_act_floats(A::prog), S4 = sp_small_floats(A::prog) / 4, SIT = (S4 + POL_THREADS - 1) / POL_THREADS;
  constexpr int D = A::prog.obs_dim, AD = A::prog.act_dim, LC = sp_leading_copies(A::prog), J0 = sp_next_dense(A::prog, 0);
  constexpr int IN_FLOATS = POL_TILE * (sp_ld(A::prog, 0) + sp_ld(A::prog, 1));  // buffers 0 and 1 take the gathered inputs
  static_assert(D == spec_obs_dim<SPEC>() && AD == 4, "policy and env agree on the row");
  float* small = lds + ACT;
  float* otile = small + S4 * 4;           // [16][D] observation rows: obs0, then what the env waves' stage 3 wrote during the last pass
  float* atile = otile + POL_TILE * D;     // [16][4] actions of the last pass (previous action of the next)
  uint8_t* trt = reinterpret_cast<uint8_t*>(atile + POL_TILE * AD);  // [16] truncated flags of the last step
  constexpr unsigned CONST_OPS = fused_const_ops<A>;
  constexpr int ZD = CONST_OPS ? A::prog.aux_dim : 0;
  constexpr int PV = fp_prev_op<A>();
  static_assert(PV >= 0 && A::prog.op[PV >= 0 ? PV : 0].in_off == 0 && A::prog.op[PV >= 0 ? PV : 0].in_dim == A::prog.act_dim,
                "one leading gather takes the whole previous action");
  constexpr int prev_base = sp_base(A::prog, A::prog.op[PV].out_buf) + A::prog.op[PV].out_off, prev_ld = sp_ld(A::prog, A::prog.op[PV].out_buf);
  float* ztile = atile + POL_TILE * AD + 16;   // [16][aux_dim] the parameter embedding of the first step
  constexpr int z_base = sp_base(A::prog, A::prog.aux_buf) + A::prog.aux_off, z_ld = sp_ld(A::prog, A::prog.aux_buf);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = tid & 63;
  const int env0 = blockIdx.x * POL_TILE, rows = min(POL_TILE, a.n - env0), n = a.n;
  const bool want_value = value != nullptr;

  if (wave < POL_WAVES) {
    // ==================================================================== waves 0..3: the network
    SCtx c;
    c.lds = lds; c.small = small; c.tid = tid; c.wave = wave; c.li = lane & 15; c.lg = lane >> 4;
    c.mir = reinterpret_cast<_Float16*>(lds + sp_base(A::prog, A::prog.n_bufs));
    c.wrs = pol_weight_rsrc(p.packed + p.wsplit_off); c.lane16 = lane * 16; c.lane = lane;
    c.obs = otile; c.prev_actions = atile; c.prev_truncated = trt;   // the gathers read LDS tiles, rows 0..rows-1
    c.n_envs = rows; c.env0 = 0; c.want_value = want_value;
    c.small_global = p.packed + p.prog_ints; c.state = nullptr; c.counter = 0u;  // feed-forward networks only: no history rings
    c.skip_ops = 0u;
    SPre<A, J0> pre;
    s_prefetch<A, J0>(c, pre);
    {  // prologue: parameters mirror, first observation / previous action, cleared activations
      const float4* src = reinterpret_cast<const float4*>(p.packed + p.prog_ints);
      float4 sm[SIT];
#pragma unroll
      for (int it = 0; it < SIT; it++) {
        const int k = tid + it * POL_THREADS;
        sm[it] = k < S4 ? src[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      for (int k = tid; k < POL_TILE * D; k += POL_THREADS) otile[k] = k < rows * D ? obs0[(size_t)env0 * D + k] : 0.f;
      for (int k = tid; k < POL_TILE * AD; k += POL_THREADS) atile[k] = (prev0 && k < rows * AD) ? prev0[(size_t)env0 * AD + k] : 0.f;
      if (tid < POL_TILE) trt[tid] = 0;
      for (int k = tid; k < ACT; k += POL_THREADS) lds[k] = 0.f;
      float4* dst = reinterpret_cast<float4*>(small);
#pragma unroll
      for (int it = 0; it < SIT; it++) {
        const int k = tid + it * POL_THREADS;
        if (k < S4) dst[k] = sm[it];
      }
    }
    __syncthreads();   // P
    constexpr int ldl = sp_ld(A::prog, A::prog.logits_buf), NL = A::prog.n_logits;
    constexpr int lg_base = sp_base(A::prog, A::prog.logits_buf) + A::prog.logits_off;
    constexpr int VB = A::prog.value_buf < 0 ? 0 : A::prog.value_buf;
    constexpr int v_base = sp_base(A::prog, VB) + A::prog.value_off, v_ld = sp_ld(A::prog, VB);
    constexpr bool has_value = A::prog.value_buf >= 0;
    for (int t = 0; t < T; t++) {
      FP_STAMP(0);
      if (t == 0) {   // the first pass gathers from the tiles the prologue filled (two phases: loads, barrier, stores)
        SLead<A, 0, LC> lead;
        lead.load(c);
        for (int k = tid; k < IN_FLOATS; k += POL_THREADS) lds[k] = 0.f;
        FP_STAMP(4);
        __syncthreads();   // G0
        FP_STAMP(5);
        lead.store(c);
      } else {
        // later passes: wave D staged buffers 0 and 1 during the last pass (the input buffers also held its outputs), the output
        // stage added the previous action; only the pass after the first still fetches the parameter embedding itself
        for (int k = tid; k < IN_FLOATS / 4; k += POL_THREADS) reinterpret_cast<float4*>(lds)[k] = reinterpret_cast<const float4*>(stage)[k];
        FP_STAMP(4);
        __syncthreads();   // G0
        FP_STAMP(5);
        if (CONST_OPS && t == 1)
          for (int k = tid; k < POL_TILE * ZD; k += POL_THREADS) lds[z_base + (k / (ZD ? ZD : 1)) * z_ld + k % (ZD ? ZD : 1)] = ztile[k];
      }
      FP_STAMP(6);
      __syncthreads();   // G1
      FP_STAMP(1);
      s_run<A, LC, J0>(c, pre);   // one barrier per executed op (fp_run_barriers)
      FP_STAMP(2);
      if (CONST_OPS && t == 0) {  // the embedding is final once the program has run: keep it for the rest of the fragment
        for (int k = tid; k < POL_TILE * ZD; k += POL_THREADS) ztile[k] = lds[z_base + (k / (ZD ? ZD : 1)) * z_ld + k % (ZD ? ZD : 1)];
        c.skip_ops = CONST_OPS;
      }
      // (the output addresses are functions of the thread index alone; left to itself the compiler computes them once, before the
      // loop, and with 256 registers keeps them in scratch: an opaque copy of the index makes it recompute them, a few instructions)
      int tid_o = tid;
      asm volatile("" : "+v"(tid_o));
      if (has_value && c.want_value && tid_o < rows) value[(size_t)t * n + env0 + tid_o] = lds[v_base + tid_o * v_ld];
      FP_STAMP(7);
      PolSample st = smp;
      st.counter = smp.counter + (unsigned int)t;
      pol_outputs(lds + lg_base, ldl, NL, AD, env0, n, tid_o, lds + ACT - POL_SCRATCH, st, actions + (size_t)t * n * AD,
                  logp ? logp + (size_t)t * n : nullptr, logits ? logits + (size_t)t * n * NL : nullptr, atile, p.dist,
                  stage + prev_base, prev_ld, trt, rows);
      // the next pass's first layer: requested behind the output stage, in flight across the barrier and the gather (requested in
      // front of it, the 32 destination registers collide with the output stage's temporaries under the 256-register cap and the
      // compiler waits for the loads right there: s_waitcnt vmcnt(1), 0.3 us)
      s_prefetch<A, J0>(c, pre);
      FP_STAMP(8);
      __syncthreads();   // O: the action is in LDS; the next pass starts at once (its inputs were ready long ago)
      FP_STAMP(3);
    }
    return;
  }

  // ====================================================================== waves 4..7: the env roles, lanes 0..15
  // (each shares its SIMD with a network wave that mostly waits for 32-cycle MFMAs: the env wave's instructions go first)
  __builtin_amdgcn_s_setprio(3);
  const int role = wave - POL_WAVES;
  const bool col = lane < POL_TILE;          // this lane has a column of the hand-over arrays
  const bool live = lane < rows;
  const int i = env0 + lane;
  const int il = live ? i : a.n - 1;         // lanes past the batch work on a copy of the last env (no stores)
  EnvRegs e;
  load_env_planes<true, false, false>(a.g, a.npad, il, e);
  float ref0[4] = {a.ref[0], a.ref[1], a.ref[2], a.ref[3]};
  if (a.ref_mode == QD_REF_STATIC && a.per_env_ref) {
    const float4 r = a.g[G_REF * a.npad + il];
    ref0[0] = r.x; ref0[1] = r.y; ref0[2] = r.z; ref0[3] = r.w;
  }
  // what the network executes per step, passed in the same order: G0 G1 | layer barriers | (log-prob reduction) | O
  const int nb_first = fp_run_barriers<A>(want_value, 0u), nb_later = fp_run_barriers<A>(want_value, CONST_OPS);
  const int extra = logp ? 1 : 0;
#define FP_PASS_REST(nb)                                      \
  do {                                                        \
    for (int k_ = 3; k_ < (nb) + extra; k_++) coop_barrier(); \
    coop_barrier(); /* O */                                   \
  } while (0)

  if (role == 0) {
    // ================================================================ wave A: factorisation, solve, integration, resets; the filter
    if (col) {
      fp_put_state(L.st, lane, e.s);
      L.info[lane] = make_uint4(0u, e.episode, (uint32_t)e.num_steps, 0u);
    }
    if (lane == 0) { L.tag = 0; L.ready = 0u; }
    coop_barrier();   // P
    // the mass matrix in qd_dynamics.h's latency arrangement (k_rollout_lat's arithmetic: per-env coefficients once, hinges first,
    // the 3 x 3 that is left by its adjugate, the solve as dot products; 60 registers across the layer barrier where the LDL^T
    // factor held 100)
    const LatConsts<double> K = lat_consts(e.M, a.h);
    Inv5<double> v5 = {};
    Rot5<double> r5 = {};
    M3<float> R;
    R.m00 = R.m11 = R.m22 = 1.f; R.m01 = R.m02 = R.m10 = R.m12 = R.m20 = R.m21 = 0.f;
    V3<float> w0 = mk<float>(0.f, 0.f, 0.f);
    // the accelerometer reading of the last solve: the explicit accelerations from the implicit ones (explicit_from_implicit)
    auto reading = [&]() {
      V3<double> a0ex;
      V3<float> angex;
      float d1, d2;
      explicit_from_implicit(K, v5, explicit_weights(K, v5), r5.fl, r5.al, r5.t1, r5.t2, &a0ex, &angex, &d1, &d2);
      const float g = float(Const::gravity);
      return accelerometer(cvt<float>(a0ex), angex, mk<float>(g * R.m20, g * R.m21, g * R.m22),
                           mk<float>(w0.x * w0.z, w0.y * w0.z, -(w0.x * w0.x + w0.y * w0.y)));
    };
    // the part of the Euler step that reads the action: the activation filter.  It runs a pass LATE, between the next pass's two
    // gather barriers -- waves B and C read the activations behind the second one -- so that the network never waits for it.
    V3<float> sc0 = mk<float>(0.f, 0.f, 0.f), scol[4] = {sc0, sc0, sc0, sc0}, acc_last = sc0;   // sens: a reset lane's affine form, the last reading
    bool pend = false;
    auto filter = [&](int ts) {   // ts: the step whose action this is
      if (col) {
        rc_filter<SPEC>(a, e.M, e.s, *reinterpret_cast<const float4*>(atile + lane * 4));
        L.st[4][lane] = make_float4(e.s.a0, e.s.a1, e.s.a2, e.s.a3);
        if (sens) {   // what the row of step ts carries of the action
          float* row = obs + ((size_t)ts * n + i) * D;
          const float av[4] = {e.s.a0, e.s.a1, e.s.a2, e.s.a3};
          fp_patch_obs<A, 0, LC, 4>(lds, lane, act_at, av);   // between the gather barriers: the stage copy is in, the first layer not started
          if (live) { row[act_at] = av[0]; row[act_at + 1] = av[1]; row[act_at + 2] = av[2]; row[act_at + 3] = av[3]; }
          if (pend) {   // the reset lane's reading at its new state, with the activations it now has
            const float rd[3] = {sc0.x + scol[0].x * e.s.a0 + scol[1].x * e.s.a1 + scol[2].x * e.s.a2 + scol[3].x * e.s.a3,
                                 sc0.y + scol[0].y * e.s.a0 + scol[1].y * e.s.a1 + scol[2].y * e.s.a2 + scol[3].y * e.s.a3,
                                 sc0.z + scol[0].z * e.s.a0 + scol[1].z * e.s.a1 + scol[2].z * e.s.a2 + scol[3].z * e.s.a3};
            acc_last = mk<float>(rd[0], rd[1], rd[2]);
            fp_patch_obs<A, 0, LC, 3>(lds, lane, acc_at, rd);
            if (live) { row[acc_at] = rd[0]; row[acc_at + 1] = rd[1]; row[acc_at + 2] = rd[2]; }
          }
        }
      }
    };
    for (int t = 0; t < T; t++) {
      coop_barrier();   // G0
      if (t > 0) filter(t - 1);   // with the action of step t - 1 (the network rewrites the tile at the end of this pass)
      coop_barrier();   // G1
      FP_STAMP(16);
      // ---------------------------------------------------------- stage 1
      if (col) {
        const Tether<float> tg = tether_geometry(e.s.th1, e.s.th2);
        v5 = mass_inverse(K, tether_hp<double>(tg.s1, tg.c1, tg.s2, tg.c2));
        rc_ref(a, i, e.num_steps, ref0, e.ref);
        // everything the solve reads exists BEFORE the barrier (the barrier is an asm the compiler moves pure arithmetic across)
        rc_pin(v5.cxx, v5.cxy, v5.cxz); rc_pin(v5.cyy, v5.cyz, v5.czz); rc_pin(v5.U1); rc_pin(v5.U2);
        rc_pin(v5.s11, v5.s12, v5.s22); rc_pin(v5.rc); rc_pin(v5.kp1); rc_pin(v5.kp2);
      }
      FP_STAMP(17);
      // Stage 2 needs waves B and C's wrenches and wave D's settled reward (it overwrites L.pre) -- not the network's first layer, whose
      // barrier is 0.4 us further on (its 256 outputs' tanh + split epilogue): the three waves count themselves in, this wave polls
      // the count and passes layer barrier 1 INSIDE stage 2.  (A wave's LDS operations execute in order: the count lands behind the data.)
      // (Rows with the accelerometer keep the barrier in front: their stage 2 opens with the reading, and with the barrier behind it
      // the network waited -- train_LSTM.py's pair at 8192 envs 9.42 against 9.18 us per step; RMA_full at 4096: 5.40 against 5.62.)
      constexpr bool EARLY = !sens;
      if (EARLY) {
        const unsigned want = 3u * (unsigned)(t + 1);
        int spins = 0;
        for (; spins < (1 << 20); spins++) {
          if (*(volatile unsigned int*)&L.ready == want) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (spins == (1 << 20) && lane == 0) health_count(a, 0);
      } else {
        coop_barrier();   // layer barrier 1
      }
      asm volatile("" ::: "memory");
      // ---------------------------------------------------------- stage 2: the solve (a), the integration and what follows it (b)
      auto stage2a = [&]() {
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
          r5 = solve_inv5_rot(v5, ap, in);
        }
        w0 = mk<float>(e.s.wx, e.s.wy, e.s.wz);
        if (sens) {   // the reading of this step (the ONLY place it is evaluated: the arena gets the same bits)
          acc_last = reading();
          L.acc[lane] = make_float4(acc_last.x, acc_last.y, acc_last.z, 0.f);
        }
      };
      auto stage2b = [&]() {
        pend = false;
        Accel<float> im;
        im.ang = cvt<float>(r5.al); im.thdd1 = (float)r5.t1; im.thdd2 = (float)r5.t2;
        im.lin = mul(R, cvt<float>(solve_inv5_lin(K, v5, r5)));
        integrate_motion<float, true>(e.s, im, a.h);
        e.flags &= ~FLAG_ACC_STALE;
        e.num_steps += 1;
        const int steps_post = e.num_steps;
        bool tr;
        {
          const float dx = e.s.px - e.ref[0], dy = e.s.py - e.ref[1], dz = e.s.pz - e.ref[2];
          const float dist = qsqrt(dx * dx + dy * dy + dz * dz);
          tr = spec_term<SPEC>(a) == QD_TERM_SIMPLE ? dist > 0.5f : (!(dist <= a.max_distance) || e.num_steps >= a.max_steps);
        }
        const bool rst = a.auto_reset && tr;
        if (rst) {
          fp_put_state(L.pre, lane, e.s);
          State<float> ns;   // the new episode's state; the activations carry over (reset_bookkeeping) and are filtered below
          sample_episode<true>(a, i, e.episode, ns);
          ns.a0 = e.s.a0; ns.a1 = e.s.a1; ns.a2 = e.s.a2; ns.a3 = e.s.a3;
          e.s = ns;
          reset_bookkeeping(e.s, e.episode, e.num_steps);
          if (sens) {
            sensor_affine_lat<float>(e.M, K, e.s, &sc0, scol);
            pend = true;
          } else {
            e.flags |= FLAG_ACC_STALE;
          }
        }
        fp_put_state(L.st, lane, e.s);
        L.info[lane] = make_uint4((tr ? 1u : 0u) | (rst ? 2u : 0u), e.episode, (uint32_t)steps_post, 0u);
        trt[lane] = tr ? 1 : 0;   // the next pass's previous-action gather (read after O)
      };
      // layer barrier 1 in the MIDDLE of stage 2 (the solve is behind, the integration ahead): about where the network arrives at it, so
      // that neither waits long for the other (in front of stage 2 this wave idled 0.4 us; behind it the network would)
      if (EARLY) {
        if (col) stage2a();
        coop_barrier();   // layer barrier 1
        if (col) stage2b();
      } else if (col) {
        stage2a();
        stage2b();
      }
      // published: wave D starts on the row at once (value, then tag -- a wave's LDS operations execute in order)
      asm volatile("" ::: "memory");
      if (lane == 0) *(volatile int*)&L.tag = t + 1;
      FP_STAMP(18);
      coop_barrier();   // layer barrier 2
      coop_barrier();   // layer barrier 3 (wave D's stage)
      FP_PASS_REST(t == 0 ? nb_first : nb_later);
      FP_STAMP(19);
    }
    filter(T - 1);   // the last step's
    // what a per-step launch leaves in the arena: the state, and the reading of the last step (quirk C-6; stale where that step reset)
    if (live) {
      e.acc = sens ? acc_last : reading();   // (v5, r5, R, w0: the last step's)
      store_env(a, i, e);
    }
  } else if (role == 1) {
    // ================================================================ wave B: thrust + drag on the three bodies
    coop_barrier();   // P
    for (int t = 0; t < T; t++) {
      coop_barrier();   // G0
      coop_barrier();   // G1
      if (col) {
        State<float> s;
        fp_get_state(L.st, lane, s);
        const Tether<float> tg = tether_geometry(s.th1, s.th2);
        M3<float> Rb;
        V3<float> vb;
        attitude_min(s, &Rb, &vb);
        const V3<float> w = mk<float>(s.wx, s.wy, s.wz);
        const float g = float(Const::gravity);
        // rotors + drag on core and link (minus the core body's inertial share: forward_lat's split), drag on the tether
        const Applied<float> a1 = applied_core_link<true>(e.M, s, w, vb, tg.s1, tg.c1, mk<float>(g * Rb.m20, g * Rb.m21, g * Rb.m22));
        const Applied<float> a2 = applied_tether(e.M, s, w, vb, tg);
        L.app[0][lane] = make_float4(a1.F.x + a2.F.x, a1.F.y + a2.F.y, a1.F.z + a2.F.z, a1.t1 + a2.t1);
        L.app[1][lane] = make_float4(a1.Tq.x + a2.Tq.x, a1.Tq.y + a2.Tq.y, a1.Tq.z + a2.Tq.z, a1.t2 + a2.t2);
        L.app[2][lane] = make_float4(Rb.m00, Rb.m01, Rb.m02, Rb.m10);
        L.app[3][lane] = make_float4(Rb.m11, Rb.m12, Rb.m20, Rb.m21);
        L.app[4][lane] = make_float4(Rb.m22, 0.f, 0.f, 0.f);
      }
      if (!sens) __builtin_amdgcn_wave_barrier();
      if (!sens && lane == 0) __hip_atomic_fetch_add(&L.ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // wave A's stage 2 may read them
      coop_barrier();   // layer barrier 1
      coop_barrier();   // layer barrier 2
      coop_barrier();   // layer barrier 3
      FP_PASS_REST(t == 0 ? nb_first : nb_later);
    }
  } else if (role == 2) {
    // ================================================================ wave C: gravity + velocity products
    coop_barrier();   // P
    for (int t = 0; t < T; t++) {
      coop_barrier();   // G0
      coop_barrier();   // G1
      if (col) {
        State<float> s;
        fp_get_state(L.st, lane, s);
        const Tether<float> tg = tether_geometry(s.th1, s.th2);
        const TetherHP<double> th = tether_hp<double>(tg.s1, tg.c1, tg.s2, tg.c2);   // the same pairs the solver wave builds its inverse from
        V3<float> gt, w;
        gravity_body(s, &gt, &w);
        const Inertial<double> in = inertial_wrench_hp<float, double, false>(e.M, s, gt, w, th.d, th.y2);   // the core body's share: wave B
        L.ine[0][lane] = make_double2(in.F.x, in.F.y);
        L.ine[1][lane] = make_double2(in.F.z, in.Tq.x);
        L.ine[2][lane] = make_double2(in.Tq.y, in.Tq.z);
        L.ine[3][lane] = make_double2(in.t1, in.t2);
      }
      if (!sens) __builtin_amdgcn_wave_barrier();
      if (!sens && lane == 0) __hip_atomic_fetch_add(&L.ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      coop_barrier();   // layer barrier 1
      coop_barrier();   // layer barrier 2
      coop_barrier();   // layer barrier 3
      FP_PASS_REST(t == 0 ? nb_first : nb_later);
    }
  } else {
    // ================================================================ wave D: the observation row; reward and flags a pass late
    coop_barrier();   // P
    float sv[33];
    M3<float> Rq;
    float ref_t[4] = {0.f, 0.f, 0.f, 0.f};
    uint4 info = make_uint4(0u, 0u, 0u, 0u);
    bool rst = false;
#pragma unroll
    for (int k = 0; k < 33; k++) sv[k] = 0.f;
    Rq.m00 = Rq.m01 = Rq.m02 = Rq.m10 = Rq.m11 = Rq.m12 = Rq.m20 = Rq.m21 = Rq.m22 = 0.f;
    // reward and flags of step `ts` from what the row's construction left in registers and the action in the tile.  Like the filter
    // it runs behind the next pass's gather barriers (the state before a truncated lane's reset is overwritten after layer barrier 1).
    auto settle = [&](int ts) {
      if (col) {
        const float4 u = *reinterpret_cast<const float4*>(atile + lane * 4);
        const float act4[4] = {u.x, u.y, u.z, u.w};
        const bool simple = spec_term<SPEC>(a) == QD_TERM_SIMPLE;
        float rw;
        if (simple) {   // SimpleDrone.step's reward on this model (env_step: 0.1 - |pos - ref|)
          const float dx = sv[0] - ref_t[0], dy = sv[1] - ref_t[1], dz = sv[2] - ref_t[2];
          rw = 0.1f - qsqrt(dx * dx + dy * dy + dz * dz);
        } else {
          rw = qd::reward<float>(spec_reward<SPEC>(a), sv, act4, (int)info.z, ref_t, a.max_distance, &Rq);
        }
        if (__any(rst ? 1 : 0)) {   // the reward of a truncated lane is of the state BEFORE its reset
          State<float> pst;
          fp_get_state(L.pre, lane, pst);
          float sv2[33];
          M3<float> Rq2;
          drone_state<float, true>(pst, mk<float>(0.f, 0.f, 0.f), ref_t, e.par, sv2, &Rq2);
          float rw2;
          if (simple) {
            const float dx = sv2[0] - ref_t[0], dy = sv2[1] - ref_t[1], dz = sv2[2] - ref_t[2];
            rw2 = 0.1f - qsqrt(dx * dx + dy * dy + dz * dz);
          } else {
            rw2 = qd::reward<float>(spec_reward<SPEC>(a), sv2, act4, (int)info.z, ref_t, a.max_distance, &Rq2);
          }
          if (rst) rw = rw2;
        }
        if (live) {
          __builtin_nontemporal_store(rw, reward + (size_t)ts * n + i);
          __builtin_nontemporal_store((uint8_t)(info.x & 1u), trunc + (size_t)ts * n + i);
        }
      }
    };
    for (int t = 0; t < T; t++) {
      coop_barrier();   // G0
      coop_barrier();   // G1
      FP_STAMP(42);
      if (t > 0) settle(t - 1);
      FP_STAMP(43);
      if (!sens) __builtin_amdgcn_wave_barrier();
      if (!sens && lane == 0) __hip_atomic_fetch_add(&L.ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // L.pre is read: wave A's stage 2 may overwrite it
      coop_barrier();   // layer barrier 1
      // s_{t+1}, the flags and (truncated lanes) the state before the reset: as soon as wave A has published them, not a barrier later
      // (wave A never waits for this wave, so the poll ends; it is bounded all the same)
      {
        int spins = 0;
        for (; spins < (1 << 20); spins++) {
          if (*(volatile int*)&L.tag == t + 1) break;
          __builtin_amdgcn_s_sleep(2);
        }
        // a poll that ran out would hand wrong rows and rewards on in silence: counted instead (qd_health_counters; the tests
        // hold the count at zero)
        if (spins == (1 << 20) && lane == 0) health_count(a, 0);
      }
      asm volatile("" ::: "memory");
      FP_STAMP(40);
      if (col) {
        EnvRegs ed;   // what write_obs_row reads of an env: its state and reference
        fp_get_state(L.st, lane, ed.s);
        info = L.info[lane];
        rst = (info.x & 2u) != 0u;
        rc_ref(a, i, (int)info.z - 1, ref0, ref_t);   // the reference the step ran with (episode step before the increment)
        ed.ref[0] = ref_t[0]; ed.ref[1] = ref_t[1]; ed.ref[2] = ref_t[2]; ed.ref[3] = ref_t[3];
        if (rst && a.ref_mode != QD_REF_STATIC) moving_reference(a, i, 0, ed.ref);   // a new episode's first row
        drone_state<float, true>(ed.s, mk<float>(0.f, 0.f, 0.f), ed.ref, e.par, sv, &Rq);
        if (live) {
          float* row = otile + lane * D;
          write_obs_row<true, SPEC>(a, ed, sv, &Rq, row);
          if (sens) {   // a reset lane's entries follow a pass later (wave A)
            const float4 x = L.acc[lane];
            row[acc_at] = rst ? 0.f : x.x; row[acc_at + 1] = rst ? 0.f : x.y; row[acc_at + 2] = rst ? 0.f : x.z;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int k = lane; k < rows * D; k += 64) {
        if (sens) {   // (not the entries wave A writes later: two waves' stores to one address have no order)
          const int r = k / D, c = k - r * D;
          if ((c >= act_at && c < act_at + 4) || (c >= acc_at && c < acc_at + 3 && (L.info[r].x & 2u))) continue;
        }
        __builtin_nontemporal_store(otile[k], obs + ((size_t)t * n + env0) * D + k);
      }
      FP_STAMP(41);
      coop_barrier();   // layer barrier 2
      // the next pass's inputs (everything but the previous action): the network has a layer or two to go, this wave is idle
      for (int k = lane; k < IN_FLOATS / 4; k += 64) reinterpret_cast<float4*>(stage)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      fp_stage_obs<A, 0, LC>(otile, stage, rows, lane);
      if (CONST_OPS && t >= 1)   // the parameter embedding of the first pass (the network stashed it behind that pass's layers)
        for (int k = lane; k < POL_TILE * ZD; k += 64) stage[z_base + (k / (ZD ? ZD : 1)) * z_ld + k % (ZD ? ZD : 1)] = ztile[k];
      FP_STAMP(44);
      coop_barrier();   // layer barrier 3
      FP_PASS_REST(t == 0 ? nb_first : nb_later);
    }
    settle(T - 1);
  }
#undef FP_PASS_REST
}

// ---- host ----
// dynamic LDS of one instantiation: activations + mirror + scratch, the small region, the observation / action tiles, flags, z stash
template <int SPEC, class A>
constexpr size_t fp_lds_bytes() {
  return ((size_t)sp_act_floats(A::prog) + sp_small_floats(A::prog) + POL_TILE * (A::prog.obs_dim + A::prog.act_dim) + 16 + POL_TILE * 8) * sizeof(float);
}

hipError_t launch_rollout_fused_pipe(int arch, const KArgs& k, const PolArgs& pa, size_t lds_bytes, int T, const PolSample& smp,
                                     const float* obs0, const float* prev0, float* obs, float* actions, float* reward, uint8_t* trunc,
                                     float* logp, float* logits, float* value, hipStream_t stream) {
  KArgs kk = k;
  kk.use_pool = 0;
  kk.main_blocks = 0;
  const dim3 grid((k.n + POL_TILE - 1) / POL_TILE), block(FP_THREADS);
  (void)hipGetLastError();
  // more dynamic LDS than the default limit: the instantiation opts in (160 KB per CU on gfx950), once per process and arch
#define FP_LAUNCH(SPECV, ARCH)                                                                                                        \
  do {                                                                                                                                \
    const size_t lds_ = lds_bytes > fp_lds_bytes<SPECV, ARCH>() ? lds_bytes : fp_lds_bytes<SPECV, ARCH>();                            \
    if (lds_ > 64 * 1024) {                                                                                                           \
      static size_t opted = 0;                                                                                                        \
      if (lds_ > opted) {                                                                                                             \
        const hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout_fused_pipe<SPECV, ARCH>),                   \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_);                             \
        if (e_ != hipSuccess) return e_;                                                                                              \
        opted = lds_;                                                                                                                 \
      }                                                                                                                               \
    }                                                                                                                                 \
    hipLaunchKernelGGL((k_rollout_fused_pipe<SPECV, ARCH>), grid, block, lds_, stream, kk, pa, T, smp, obs0, prev0, obs, actions,     \
                       reward, trunc, logp, logits, value);                                                                           \
  } while (0)
  switch (arch) {
    case 1: FP_LAUNCH(SPEC_RMA, ArchRmaFull); break;
    case 2: FP_LAUNCH(SPEC_RMA, ArchRmaModel); break;
    case 7: FP_LAUNCH(SPEC_RMA, ArchCustomMlp); break;
    case 9: FP_LAUNCH(SPEC_RMA, ArchRmaSmaller); break;
    case 5: FP_LAUNCH(SPEC_LSTM, ArchCnnEst); break;   // train_LSTM.py: the 23-value rows with the accelerometer
    default: return hipErrorNotSupported;
  }
#undef FP_LAUNCH
  return hipGetLastError();
}

}  // namespace qd

#if QD_POL_TILE == 32
// qd_rollout_fused32.hip compiles this file a second time with 32 envs per workgroup, in a namespace of its own; the library's
// host side (qd_policy_host.inc) reaches that copy through this plain entry point.  KArgs / PolArgs / PolSample are the same
// structs in both copies.
extern "C" int qd_fused32_launch(int arch, const void* k, const void* pa, int T, const void* smp, const float* obs0, const float* prev0,
                                 float* obs, float* actions, float* reward, uint8_t* trunc, float* logp, float* logits, float* value,
                                 void* stream) {
  return (int)qd::launch_rollout_fused_pipe(arch, *static_cast<const qd::KArgs*>(k), *static_cast<const qd::PolArgs*>(pa), 0, T,
                                            *static_cast<const qd::PolSample*>(smp), obs0, prev0, obs, actions, reward, trunc, logp,
                                            logits, value, static_cast<hipStream_t>(stream));
}
#endif

#if defined(QD_STAMPS) && QD_POL_TILE == 16
extern "C" int qd_debug_read_fpstamps(unsigned long long* out_host) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(qd::qd_fpstamps), sizeof(qd::qd_fpstamps));
}
#endif
