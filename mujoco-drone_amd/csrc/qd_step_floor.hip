// qd_step_floor.hip -- k_step_floor: the env step of configurations with the floor plane (qd_config.floor_contact; SURVEY 8f-1,
// env_gen.py:97), with the contact solve as a lane-group computation (qd_contact_group.h).
//
// A workgroup of four wavefronts steps 32 envs.  Per substep the lower half of the first wavefront runs the unconstrained forward
// dynamics, one env per lane; the lanes whose env can reach the floor (height test) describe it in LDS, compacted; then the four
// wavefronts solve those contact problems side by side, 8 envs per wavefront with 8 LANES PER ENV (cross-lane reductions of
// the gradient and of J^T D J, qd_contact_group.h); then the owning lanes integrate.  In flight the contact code is one ballot
// and two workgroup barriers; with every env on the floor the 4096-env batch is 512 wavefronts' worth of solves on 128 CUs
// instead of 64 wavefronts grinding through 64 scratch-resident solves each.  Replaces k_step<LOAD, BLOCK, SPEC_FLOOR>, whose per-lane solve (qd_contact.h, kept as the host twin's and the
// tests' reference implementation) needed 1.8 KB of scratch per lane and 155 us per step of 4096 load-model envs in contact.
#include "qd_contact_group.h"
#include "qd_env_device.h"

namespace qd {

constexpr int SF_THREADS = 256;   // four wavefronts per workgroup of CG_BLOCK_ENVS = 32 envs

// what the owning lane keeps of its env across the substeps -- parked in LDS while the wavefront works on the contact problem,
// whose Newton iterations want the registers (the allocator otherwise keeps these ~110 values alive through the solve: 285 spills)
template <bool LOAD>
struct SfOwn {
  EnvRegs e;
  Accel<float> ex, im;
  V3<float> acc, lin0, ang0;
  float c0, c1, c2, c3;
  double fc[FLOOR_CONSTS];   // the parameter set's floor constants (arena planes behind the raw parameters)
};

template <bool LOAD>
__global__ __launch_bounds__(SF_THREADS) void k_step_floor(KArgs a, const float* __restrict__ actions, float* __restrict__ obs,
                                                           float* __restrict__ reward, uint8_t* __restrict__ trunc) {
  __shared__ float tile[CG_BLOCK_ENVS * QD_MAX_OBS];
  __shared__ CgLds G;
  __shared__ SfOwn<LOAD> park[CG_BLOCK_ENVS];
  __shared__ int n_touch;
  if ((int)blockIdx.x >= a.main_blocks) {   // sampler workgroup (reset pool, qd_env_device.h): 64 envs per wavefront
    sampler_wave<LOAD>(a, ((int)blockIdx.x - a.main_blocks) * SF_THREADS + threadIdx.x);
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int base_env = blockIdx.x * CG_BLOCK_ENVS;
  const int i = base_env + lane;
  const bool owner = wave == 0 && lane < CG_BLOCK_ENVS;   // one env per lane of the first wavefront's lower half
  const bool live = owner && i < a.n;
  const int il = i < a.n ? i : a.n - 1;
  SfOwn<LOAD> o;
  float4 action = make_float4(0.f, 0.f, 0.f, 0.f);
  if (owner) {
    load_env<LOAD, false, false>(a, il, o.e);
    action = reinterpret_cast<const float4*>(actions)[il];
    float c0 = action.x, c1 = action.y, c2 = action.z, c3 = action.w;
    if (a.ctrl_map == QD_CTRL_AFFINE) { c0 = 0.1f + 0.9f * c0; c1 = 0.1f + 0.9f * c1; c2 = 0.1f + 0.9f * c2; c3 = 0.1f + 0.9f * c3; }
    o.c0 = qclamp(c0, 0.f, 1.f); o.c1 = qclamp(c1, 0.f, 1.f); o.c2 = qclamp(c2, 0.f, 1.f); o.c3 = qclamp(c3, 0.f, 1.f);
    // sizes, reach and body_invweight0 of this parameter set: derived once per parameter set by k_floor_consts
#pragma unroll
    for (int k = 0; k < FLOOR_CONSTS; k++) o.fc[k] = a.raw[(size_t)(RAW_PARAMS + k) * a.npad + il];
  }
  for (int k = 0; k < a.frame_skip; k++) {
    bool touch = false;
    int rank = 0;
    if (wave == 0) SF_STAMP(0);
    if (wave == 0) {
      if (owner) {
        forward<float, LOAD>(o.e.M, o.e.s, a.h, &o.ex, &o.im, &o.acc);
        o.lin0 = o.ex.lin; o.ang0 = o.ex.ang;
        touch = live && !((double)o.e.s.pz > o.fc[FC_REACH]);
      }
      SF_STAMP(1);
      const unsigned long long mask = __ballot(touch ? 1 : 0);
      rank = __popcll(mask & ((1ull << lane) - 1ull));
      if (lane == 0) n_touch = __popcll(mask);
      if (mask != 0ull) {   // (wave-uniform) somebody can reach the floor: describe those envs, park everybody's registers
        if (touch) cg_publish<LOAD>(G.rec[rank], o.e.M, o.e.s, o.fc, o.ex);
        if (owner) park[lane] = o;
      }
      SF_STAMP(2);
    }
    __syncthreads();
    if (wave == 0) SF_STAMP(3);
    const int total = n_touch;
    if (total > 0) {   // (workgroup-uniform) the four wavefronts take 8 touching envs each, 8 lanes per env
      for (int base = wave * CG_ENVS; base < total; base += 4 * CG_ENVS) cg_solve<LOAD>(G, G.w[wave], base, total, (double)a.h);
      if (wave == 0) SF_STAMP(4);
      __syncthreads();
      if (wave == 0) SF_STAMP(5);
      if (owner) {
        o = park[lane];
        double fz = 0.0;
        if (touch) cg_collect<LOAD>(G.res[rank], o.ex, o.im, &fz);
        if (fz > 0.0) {
          // the accelerometer with the constrained accelerations (MuJoCo evaluates acceleration sensors after the constraint solve):
          // acc = R^T lin + g~ + alpha x r_s + w x (w x r_s) at the site r_s = (0, 0, sense_z): only the first and third term changed
          const State<float>& s = o.e.s;
          const float qn = frsq(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);
          const M3<float> R = quat2mat(s.qw * qn, s.qx * qn, s.qy * qn, s.qz * qn);
          const V3<float> dl = mulT(R, mk<float>(o.ex.lin.x - o.lin0.x, o.ex.lin.y - o.lin0.y, o.ex.lin.z - o.lin0.z));
          const float sz = float(Const::sense_z), dax = o.ex.ang.x - o.ang0.x, day = o.ex.ang.y - o.ang0.y;
          o.acc = o.acc + dl + mk<float>(day * sz, -dax * sz, 0.f);
        }
      }
    }
    if (wave == 0) SF_STAMP(6);
    if (owner) {
      integrate<float, LOAD>(o.e.M, o.e.s, o.im, o.c0, o.c1, o.c2, o.c3, a.h);
      o.e.acc = o.acc;
    }
    if (wave == 0) SF_STAMP(7);
    __syncthreads();   // (n_touch and the work areas are rewritten by the next substep)
  }
  if (wave != 0) return;
  // ---- the rest of the step, as env_step (qd_kernels.hip) with run-time dispatch
  EnvRegs& e = o.e;
  if (owner) {
    e.flags &= ~FLAG_ACC_STALE;
    e.num_steps += 1;
    float sv[33];
    const float act4[4] = {action.x, action.y, action.z, action.w};
    M3<float> Rq;
    bool tr;
    float r;
    if (a.term_kind == QD_TERM_SIMPLE) {   // SimpleDrone.step: terminated = |pos - ref| > 0.5, reward = 0.1 - |pos - ref| (SimpleDrone.py:57-60)
      const float dx = e.s.px - e.ref[0], dy = e.s.py - e.ref[1], dz = e.s.pz - e.ref[2];
      const float d = qsqrt(dx * dx + dy * dy + dz * dz);
      tr = d > 0.5f;
      r = 0.1f - d;
    } else {
      drone_state<float, LOAD>(e.s, e.acc, e.ref, e.par, sv, &Rq);
      tr = truncated<float>(sv, e.ref, e.num_steps, a.max_distance, a.max_steps);
      r = qd::reward<float>(a.reward_kind, sv, act4, e.num_steps, e.ref, a.max_distance, &Rq);
    }
    if (a.auto_reset && tr) {
      if (live) reset_in_step<LOAD, true>(a, i, e);
      if (a.ref_mode != QD_REF_STATIC) moving_reference(a, i, 0, e.ref);
      if (a.term_kind != QD_TERM_SIMPLE) drone_state<float, LOAD>(e.s, e.acc, e.ref, e.par, sv, &Rq);
    }
    write_obs_row<LOAD, SPEC_FLOOR>(a, e, sv, &Rq, tile + lane * a.D);
    if (live) {
      store_env(a, i, e);
      __builtin_nontemporal_store(r, reward + i);
      __builtin_nontemporal_store((uint8_t)(tr ? 1 : 0), trunc + i);
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (base_env < a.n) flush_obs(tile, obs + (size_t)base_env * a.D, min(CG_BLOCK_ENVS, a.n - base_env), a.D);
}

// the floor constants of every env's current parameter set, into the arena planes behind the raw parameters
template <bool LOAD>
__global__ __launch_bounds__(64) void k_floor_consts(KArgs a) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.n) return;
  EnvRegs e;
  load_env_planes<LOAD, false, false>(a.g, a.npad, i, e);
  double fc[FLOOR_CONSTS];
  cg_floor_consts<LOAD>(e.M, a.raw[(size_t)1 * a.npad + i], LOAD ? a.raw[(size_t)4 * a.npad + i] : 0.0, LOAD ? a.raw[(size_t)5 * a.npad + i] : 0.0, fc);
#pragma unroll
  for (int k = 0; k < FLOOR_CONSTS; k++) a.raw[(size_t)(RAW_PARAMS + k) * a.npad + i] = fc[k];
}

hipError_t launch_floor_consts(const KArgs& k, bool load, hipStream_t stream) {
  const dim3 grid((k.n + 63) / 64), block(64);
  (void)hipGetLastError();
  if (load) hipLaunchKernelGGL((k_floor_consts<true>), grid, block, 0, stream, k);
  else hipLaunchKernelGGL((k_floor_consts<false>), grid, block, 0, stream, k);
  return hipGetLastError();
}

hipError_t launch_step_floor(const KArgs& k, bool load, const float* actions, float* obs, float* reward, uint8_t* trunc, hipStream_t stream) {
  KArgs kk = k;
  kk.main_blocks = (k.n + CG_BLOCK_ENVS - 1) / CG_BLOCK_ENVS;
  const dim3 grid(kk.main_blocks + (k.use_pool ? (k.n + SF_THREADS - 1) / SF_THREADS : 0)), block(SF_THREADS);
  (void)hipGetLastError();
  if (load) hipLaunchKernelGGL((k_step_floor<true>), grid, block, 0, stream, kk, actions, obs, reward, trunc);
  else hipLaunchKernelGGL((k_step_floor<false>), grid, block, 0, stream, kk, actions, obs, reward, trunc);
  return hipGetLastError();
}

#ifdef QD_STAMPS
extern "C" int qd_debug_read_sfstamps(unsigned long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(qd_sfstamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -4;
}
#endif

}  // namespace qd
