/*
 * qd.h -- C ABI of the MI355X-native vectorised quadrotor(+hanging load) environment.
 *
 * The reference (TichyTech/mujoco-drone) has no FFI: its hot path is Python that calls
 * the MuJoCo C library through the `mujoco` bindings.  This header is the boundary a
 * maintainer binds instead (ctypes stub: INTEGRATION.md); every entry point names the
 * reference interface it replaces (paths relative to the reference repo).
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary: every call returns a
 *     qd_status (0 = ok, < 0 = error) and qd_last_error() gives the message of the last
 *     failure on the calling thread;
 *   - all array arguments are DEVICE pointers owned by the caller (e.g. PyTorch-ROCm
 *     tensors' data_ptr()) unless the name ends in _host; row-major; float32 unless
 *     stated otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are
 *     asynchronous with respect to the host and ordered on that stream;
 *   - one qd_env is not thread-safe; distinct handles are independent;
 *   - simulator state lives in a caller-allocated arena of qd_arena_bytes() bytes
 *     (struct-of-float4-groups), so the library never allocates device
 *     memory and never synchronises the device.
 */
#ifndef QD_H
#define QD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QD_VERSION 3

typedef struct qd_env qd_env;

typedef enum qd_status {
  QD_OK = 0,
  QD_ERR_INVALID = -1,      /* bad argument / configuration */
  QD_ERR_SHAPE = -2,        /* "Action dimension mismatch" (mujoco_env_custom.py:200-201) */
  QD_ERR_UNSUPPORTED = -3,  /* e.g. the observation variant that raises NameError in the reference */
  QD_ERR_HIP = -4,          /* a HIP runtime call failed */
  QD_ERR_ARENA = -5,        /* arena too small or misaligned */
  QD_ERR_INDEX = -6         /* env index out of range (AssertionError in BaseDroneEnv.py:338) */
} qd_status;

enum { QD_MODEL_NOLOAD = 0, QD_MODEL_LOAD = 1 };            /* config['pendulum'] */
enum { QD_CTRL_DIRECT = 0,  /* ctrl = action            (SimpleDrone.py:55)      */
       QD_CTRL_AFFINE = 1   /* ctrl = 0.1 + 0.9*action  (BaseDroneEnv.py:269)    */ };
enum { QD_TERM_DEFAULT = 0, /* default_termination_fcn  (BaseDroneEnv.py:12-16)  */
       QD_TERM_SIMPLE = 1   /* |pos - ref| > 0.5        (SimpleDrone.py:57)      */ };

enum { QD_START_FIXED = 0,  /* random_start_pos = False   (BaseDroneEnv.py:245-256) */
       QD_START_RANDOM = 1, /* random_start_pos = True    (BaseDroneEnv.py:220-244) */
       QD_START_SIMPLE = 2  /* SimpleDrone.reset_model    (SimpleDrone.py:63-72)    */ };

enum { QD_REF_STATIC = 0, QD_REF_CIRCLE = 1, QD_REF_STEP = 2, QD_REF_RAMP = 3 };

/* observation variants: BaseDroneEnv._get_obs and the classes of observation_wrappers.py, in file order */
enum {
  QD_OBS_RAW = 0, QD_OBS_GLOBAL_RPY, QD_OBS_LOCAL_PRY, QD_OBS_FULLSTATE, QD_OBS_FULLSTATE_ZVEC, QD_OBS_PRY_ACC,
  QD_OBS_PRY_PARAMS, QD_OBS_PRY_ACC_PARAMS, QD_OBS_RPY_PARAMS, QD_OBS_RPY_FAKEPARAMS, QD_OBS_LOCAL_RPY,
  QD_OBS_PRY_ACC_NOPEND, QD_OBS_PRY_ACC_PARAMS_NOPEND /* unsupported: raises in the reference */,
  QD_OBS_RM_PARAMS, QD_OBS_ZVEC, QD_OBS_SIMPLE /* SimpleDrone._get_obs */, QD_OBS_COUNT
};
/* reward functions of rewards.py in file order, then SimpleDrone.step's */
enum {
  QD_REW_DEFAULT = 0, QD_REW_DISTANCE, QD_REW_DISTANCE_ENERGY, QD_REW_PEND_ANGLE, QD_REW_PEND_ANGLE2,
  QD_REW_PEND_ANGLE3, QD_REW_PEND_EN, QD_REW_PEND_EN2, QD_REW_PEND_EN3, QD_REW_PEND_EN4,
  QD_REW_DISTANCE_TIME_ENERGY, QD_REW_REWARD_1, QD_REW_PEND_DIST, QD_REW_PEND_DIST_HEADING, QD_REW_REWARD_2,
  QD_REW_REWARD_2_PENERGY, QD_REW_REWARD_3, QD_REW_SIMPLE, QD_REW_COUNT
};
/* attitude conversions of transformation.py */
enum { QD_TF_QUAT2RPY = 0, QD_TF_RPY2QUAT, QD_TF_QUAT2DCM, QD_TF_DCM2QUAT, QD_TF_PENDRP2QUAT };

/* Mirrors the keys BaseDroneEnv.__init__ reads (BaseDroneEnv.py:60-106); variances are
 * already multiplied by state_difficulty as at BaseDroneEnv.py:101-106. */
typedef struct qd_config {
  int32_t  num_envs;            /* config['num_drones'] */
  int32_t  model;               /* QD_MODEL_* */
  int32_t  obs_kind, reward_kind;
  int32_t  frame_skip;          /* config['skip_steps']; SimpleDrone uses 2 */
  int32_t  max_steps;
  int32_t  ctrl_map, term_kind;
  int32_t  random_start;        /* QD_START_*: config['random_start_pos'] */
  int32_t  random_params;       /* config['random_params'] */
  int32_t  auto_reset;          /* extension: re-sample truncated envs inside the step kernel */
  int32_t  per_env_reference;   /* extension: reference[N,4] lives in the arena (moving waypoints) */
  double   timestep;            /* 1 / config['frequency'] */
  double   max_distance;
  double   reference[4];        /* x, y, z, yaw */
  double   start_pos[4];
  double   max_pos_offset;
  double   angle_var[2], vel_var[3], ang_vel_var[3], pend_rp_var[2], pend_vel_var[2];
  double   param_center[6], param_width[6]; /* mass, arm_len, motor_force, motor_tau, pendulum_len, weight_mass */
  double   param_difficulty;
  uint64_t seed;
  /* extension (BASELINE config 5): moving waypoint computed inside the step kernel.
   * QD_REF_CIRCLE: gen_circle_trajectory (evaluation.py:135-138) around `reference`, phase-shifted per env:
   *   ref_i(k) = reference + (r cos(2 pi f k dt + 2 pi i/N), r sin(...), 0, 0),  k = the env's episode step,
   *   dt = frame_skip * timestep; the step taken from episode step k is rewarded against ref_i(k). */
  int32_t  ref_mode;            /* QD_REF_* */
  int32_t  floor_contact;       /* extension switch for SURVEY 8f-1: 0 = no floor (the BASELINE configurations never reach it: flight at
                                 * z = 15 m, truncation at 4 m); 1 = the drone's geoms -- the airframe's 14, with the load also the link
                                 * sphere, the tether rod and the load box -- collide with the floor plane z = 0 (env_gen.py:97) in the
                                 * manner of MuJoCo's soft contacts: pyramidal friction cone, mu = 1, default solref / solimp.
                                 * PARITY UNPINNED beyond the rest of the physics: see DESIGN.md section 4 ("Floor contact"). */
  double   ref_radius, ref_frequency;
  /* QD_REF_STEP / QD_REF_RAMP: gen_step_trajectory / gen_ramp_trajectory (evaluation.py:141-152) with
   * start_pos = `reference`, end_pos = ref_end, sampled once per env step at t_k = k dt like the reference's
   * t = arange(0, duration, 0.01):  step: t_k < ref_t0 ? reference : ref_end;
   * ramp: t_k < ref_t0 ? reference : reference + (t_k - ref_t0)/(ref_duration - ref_t0) (ref_end - reference);
   * beyond the last sample (k >= ceil(ref_duration/dt)) the last waypoint is held (extension). */
  double   ref_t0, ref_duration;
  double   ref_end[4];
} qd_config;

const char* qd_last_error(void);
int         qd_version(void);
/* SHA-256 (hex) over the sources this library was compiled from (the .hip / .h / .inc files of csrc and include/qd.h), as computed by
 * mujoco-drone_amd/build.py: lets a host check that a prebuilt libqd.so matches the tree it sits in ("" if built by hand) */
const char* qd_source_hash(void);

/* observation length D for (variant, model): observation_space.shape[0] as actually emitted */
int    qd_obs_dim(int obs_kind, int model);
/* length of BaseDroneEnv.get_drone_states() rows: 33 with the load, 29 without */
int    qd_state_dim(int model);
size_t qd_arena_bytes(int num_envs);

/* BaseDroneEnv.__init__ / SimpleDrone.__init__: bind a configuration to a caller-owned,
 * 256-byte aligned device arena.  The arena content is initialised by qd_init. */
int qd_create(const qd_config* cfg, void* arena, size_t arena_bytes, qd_env** out);
int qd_destroy(qd_env* env);
/* generate_drone_params + make_sim + mjcf_to_mjmodel + MjData() (BaseDroneEnv.py:117,125;
 * mujoco_env_custom.py:189): parameters (randomised on device if random_params), derived model
 * constants, spawn-grid qpos0 (env_gen.py:116-124), zero velocities / activations / sensor. */
int qd_init(qd_env* env, void* stream);

/* mj_resetData (mujoco_vecenv.py:393-394, reached through env.reset()): qpos0, zero velocities,
 * activations and sensor data.  Episode counters and references are kept. */
int qd_reset_data(qd_env* env, void* stream);

/* env.reference = [...] (evaluation.py:48,66) */
int qd_set_reference(qd_env* env, const double ref_host[4]);
/* evaluate_trajectory (evaluation.py:38-72): a waypoint list for the next qd_rollout_policy calls -- step k of a rollout
 * runs (and is rewarded) under traj_host[k] (the last waypoint is held past the end), while its action was computed from
 * the observation under the previous one, as `env.reference = x` before `vector_step` gives.  T = 0 clears the schedule. */
int qd_set_reference_schedule(qd_env* env, const double* traj_host, int T);
/* extension for per-env moving waypoints (BASELINE config 5): ref[N,4] */
int qd_set_reference_per_env(qd_env* env, const float* ref, void* stream);

/* reset_model(regen=True) first half (BaseDroneEnv.py:298-310): new parameters for every env,
 * new model constants, fresh MjData (activations and sensor zeroed). */
int qd_randomize_params(qd_env* env, void* stream);
/* explicit parameters, float64 raw[N,6] (what env.drone_params holds) */
int qd_set_params(qd_env* env, const double* raw, void* stream);
int qd_get_params(qd_env* env, double* raw, void* stream);

/* reset_model second half / vector_reset (BaseDroneEnv.py:312-332): sample_state for every env
 * (mask == NULL) or for envs with mask[i] != 0, num_steps = 0, mj_forward (sensor refresh);
 * activations persist (reference quirk).  obs (nullable) receives _get_obs() for ALL envs. */
int qd_reset(qd_env* env, const uint8_t* mask, float* obs, void* stream);
/* reset_at(index) (BaseDroneEnv.py:334-351) */
int qd_reset_at(qd_env* env, int index, void* stream);

/* set_state(qpos, qvel) (mujoco_vecenv.py:396-402): qpos[N,nq], qvel[N,nv], then mj_forward.
 * act (nullable, [N,4]) additionally overwrites the activations (data.act[:] = ...). */
int qd_set_state(qd_env* env, const float* qpos, const float* qvel, const float* act, void* stream);
/* data.qpos / data.qvel / data.act / data.sensordata / num_steps; any pointer may be NULL */
int qd_get_state(qd_env* env, float* qpos, float* qvel, float* act, float* sensordata, int32_t* num_steps,
                 void* stream);

/* vector_step (BaseDroneEnv.py:259-294) / SimpleDrone.step (SimpleDrone.py:54-61) for all envs:
 * ctrl map, frame_skip x mj_step, num_steps += 1, truncation, reward, observation.
 * n_action_values must equal 4*num_envs, else QD_ERR_SHAPE.  actions[N,4], obs[N,D],
 * reward[N], truncated[N] (uint8).  With auto_reset, truncated envs are re-sampled in the same
 * launch and their obs row is the first observation of the new episode.
 * Launch variants are chosen by batch size (results per env do not depend on the variant beyond float32 rounding, and not
 * at all within one): the three-wavefront cooperative kernel for train_PPO.py / train_RMA.py's configuration up to 24576
 * envs (environment variable QD_COOP_MAX_ENVS, 0 = never), one wavefront per 64 envs up to 98303, 256-thread workgroups
 * above (QD_BLOCK_THRESHOLD); reset-pool sampler workgroups ride along below 32768 envs (QD_POOL_MAX_ENVS).  The variables
 * are read once per process and exist for measurements. */
int qd_step(qd_env* env, const float* actions, int64_t n_action_values, float* obs, float* reward,
            uint8_t* truncated, void* stream);
/* T consecutive vector_steps issued by ONE call: actions[T,N,4] -> obs[T,N,D], reward[T,N], truncated[T,N]; what the sampler
 * does per rollout fragment (T consecutive BaseDroneEnv.vector_step calls, BaseDroneEnv.py:259-294; 1024-step fragments,
 * train_RMA.py:63).  For fragments whose actions are already on the device (replays, or a policy that wrote the whole
 * fragment).  Two ways of running it, chosen per env configuration (qd_fragment_kernel_name tells which):
 *   - ONE persistent launch (k_rollout_coop, csrc/qd_rollout_coop.hip) for the load model with skip_steps = 1 (every observation
 *     variant and reward; compile-time specialisations for train_PPO.py / train_RMA.py's and train_LSTM.py's configurations), at
 *     every batch size: four wavefronts per 64 envs keep the state on their CU for all T steps, only actions come in and rows /
 *     rewards / flags go out.  Same results as T x
 *     qd_step up to the rounding of two compilations of the same arithmetic (the truncation flags exactly; bit-identical to
 *     itself whatever the batch size and wherever a run is cut into fragments).  QD_OPT_PERSISTENT_FRAGMENTS = 0 (qd_set_option)
 *     or QD_PERSISTENT=0 in the environment selects the other way;
 *   - T qd_step launches (same kernels, same results, bit for bit).  Runs of >= 128 steps (QD_GRAPH_MIN_STEPS) go out as ONE
 *     HIP graph: the first call with a given (T, buffers) captures the launches, later calls replay them, so the per-step host
 *     launch path (~4-5 us, the bound of qd_step at 4096 envs) is paid once per fragment.  A shorter run is issued launch by
 *     launch the first time its (T, buffers) are seen, captured the second time and replayed from then on (a capture costs
 *     ~20 us per node, a replay 10-16 us).  Up to 8 graphs are kept per env (least recently used replaced); QD_GRAPH_MIN_STEPS
 *     >= 2^30 switches graphs off.  Graphs are captured on a stream owned by the env and replayed in `stream`;
 *     qd_set_reference invalidates them. */
int qd_step_fragment(qd_env* env, const float* actions, int T, float* obs, float* reward, uint8_t* truncated, void* stream);
/* Per-env switches of launch variants (measurements and A/B tests; results do not depend on them beyond rounding).
 *   QD_OPT_PERSISTENT_FRAGMENTS  1 (default): qd_step_fragment / qd_rollout may run as one persistent launch; 0: never.
 *   QD_OPT_LATENCY_KERNEL        1 (default): persistent fragments of at most 16384 envs (256 workgroups, one per CU) of the
 *                                load model (one substep per step, any observation variant and reward) run k_rollout_lat (csrc/qd_rollout_lat.hip), whose four wavefronts split
 *                                the step for the shortest dependent chain; 0: k_rollout_coop at every size (the variant batches
 *                                above 16384 envs always run: a comparison across the size switch that must be bit-exact sets 0). */
enum { QD_OPT_PERSISTENT_FRAGMENTS = 0, QD_OPT_LATENCY_KERNEL = 1, QD_OPT_COUNT };
int qd_set_option(qd_env* env, int option, int value);
/* The kernel that a qd_step / a qd_step_fragment of this env launches right now (static strings; the variant selector's own
 * answer, for benchmark lines and profiles: "qd::k_step_coop<1>", "qd::k_rollout_coop<1,2>", ...). */
const char* qd_step_kernel_name(const qd_env* env);
const char* qd_fragment_kernel_name(const qd_env* env);
/* How the in-kernel resets (auto_reset) since qd_init got their new state: counters[0] = served by the reset pool (an entry
 * pre-sampled by the sampler workgroups of earlier launches), counters[1] = sampled inline by the truncating lane (pool off,
 * or no entry yet).  Same results either way; inline sampling is what the pool exists to keep off the step's critical path.
 * counters: 2 x uint32 in device memory, written in stream order. */
int qd_pool_counters(qd_env* env, uint32_t* counters, void* stream);
/* Events that must not happen, counted since qd_init: counters[0] = hand-overs inside a kernel whose bounded poll ran out (the closed
 * policy loop's row wave waits for the solver wave's publication; a run-out would mean wrong rows handed on).  Zero on a healthy
 * run; the GPU tests assert it.  counters: QD_HEALTH_COUNTERS x uint32 in device memory, written in stream order. */
#define QD_HEALTH_COUNTERS 1
int qd_health_counters(qd_env* env, uint32_t* counters, void* stream);
/* T consecutive steps in ONE launch with the state held on the chip: actions[T,N,4] ->
 * obs[T,N,D], reward[T,N], truncated[T,N].  Same results as T qd_step calls (up to rounding between kernels).  The persistent
 * kernel of qd_step_fragment where that applies, else k_rollout (one wavefront per 64 envs, state in registers). */
int qd_rollout(qd_env* env, const float* actions, int T, float* obs, float* reward, uint8_t* truncated,
               void* stream);

/* ---- the analytic cascaded PID as an on-device action source ----------------
 * models/Analytic/PositionController.py:6-34 + AttitudeController.py:7-55, wired as
 * attitude_test.py:26-47: masses = mass + weight_mass + 0.2*pendulum_len, forces = motor_force,
 * inputs = entries 0:6 of the drone state vector, target = the env's (per-env / moving) reference,
 * env action = clip(ctrl - 0.1, 0, 1).  The controller memory (integrators, previous errors,
 * first-step flags) lives in the arena, one controller pair per env.
 * qd_pid_reset   : fresh PositionController / AttittudeController objects (mask[N] nullable = all).
 *                  qd_init does this once; env resets do NOT (the reference's objects are separate),
 *                  except for envs auto-reset inside qd_rollout_pid.
 * qd_pid_action  : one controller evaluation on the current state -> actions[N,4]; memory advances.
 * qd_rollout_pid : T closed-loop steps (controller -> vector_step) in ONE launch: obs[T,N,D],
 *                  reward[T,N], truncated[T,N], actions_out[T,N,4] (nullable).  Same results as
 *                  T x (qd_pid_action, qd_step).  QD_ERR_UNSUPPORTED for SimpleDrone configurations.
 *                  Envs with floor_contact, and load-model envs outside the persistent kernel (several substeps
 *                  per step, QD_OPT_PERSISTENT_FRAGMENTS = 0), run the same loop launch by launch and require
 *                  actions_out as the buffer between controller and step (QD_ERR_INVALID without it);
 *                  qd_rollout steps such envs launch by launch too. */
int qd_pid_reset(qd_env* env, const uint8_t* mask, void* stream);
int qd_pid_action(qd_env* env, float* actions, void* stream);
int qd_rollout_pid(qd_env* env, int T, float* obs, float* reward, uint8_t* truncated, float* actions_out,
                   void* stream);

/* ---- on-device policy inference -----------------------------------------------
 * The reference's actor / critic networks (models/PPO/RMA/RMA_model.py: RMA_full with
 * train_adaptation=False, RMA_model; models/PPO/SimpleMLP/SimpleMLP.py: SimpleMLPmodel; eval mode)
 * and its action distribution's deterministic sample (distributions.py:8-26, MyBetaDist) for a whole
 * env batch in one launch, float32 on the matrix cores.  A policy is a short program over up to 4
 * per-env activation buffers:
 *   QD_POL_COPY_OBS  : buf[out_buf][out_off : +in_dim] = obs row [in_off : +in_dim]
 *   QD_POL_COPY_PREV : same from the previous action (zeros where prev_truncated != 0 or no previous
 *                      action exists: RLlib's ViewRequirement(shift=-1) at episode starts)
 *   QD_POL_DENSE     : out slice = act(W in_slice + b), W float32 row-major [out_dim][in_dim] at
 *                      weights[w_off], b at weights[b_off] (torch.nn.Linear layout)
 *   QD_POL_AFFINE    : out slice = out slice * weights[w_off + c] + weights[b_off + c]
 *                      (eval-mode BatchNorm1d: scale = gamma / sqrt(var + eps), shift = beta - mean * scale)
 *   QD_POL_RING_LOAD : buf[out_buf][out_off : + rows*width] = the slots of ring `in_buf` written by the last `rows` steps
 *                      (of this step's bank), oldest first; the fill values for envs at an episode start
 *   QD_POL_RING_PUSH : this step's slot of ring `out_buf` = buf[in_buf][in_off : + width]; at an episode start every
 *                      other slot of the ring is set to the fill values
 *   QD_POL_LSTM_CELL : torch.nn.LSTM's cell update for one step: gates = in slice [4H] in the order (i, f, g, o), H = out_dim;
 *                      out slice [H] <- h' = sigmoid(o) tanh(c'), the H floats right after it hold c and are updated in
 *                      place, c' = sigmoid(f) c + sigmoid(i) tanh(g).  (The gates come from a DENSE op over [x | h] with
 *                      the weights [W_ih | W_hh] and bias b_ih + b_hh; h and c travel between steps in rings of one row.)
 * (ring ops need the step `counter` of qd_policy_act: consecutive calls must pass consecutive counters)
 * Outputs: logits[N, n_logits] (n_logits = 2 * act_dim), value[N] (if the program has a value slot) and
 * actions[N, act_dim] = alpha / (alpha + beta) with (alpha, beta) = softplus(clamp(logits, +-50)) + 1.
 * The caller owns the device buffer the packed weights live in (qd_policy_packed_bytes). */
enum { QD_POL_DENSE = 0, QD_POL_AFFINE = 1, QD_POL_COPY_OBS = 2, QD_POL_COPY_PREV = 3, QD_POL_RING_LOAD = 4, QD_POL_RING_PUSH = 5,
       QD_POL_LSTM_CELL = 6 };
enum { QD_ACT_NONE = 0, QD_ACT_TANH = 1, QD_ACT_RELU = 2 };
enum { QD_POL_VALUE_ONLY = 1 };
/* action distributions of distributions.py on the network's 2 * act_dim outputs */
enum { QD_DIST_BETA = 0,              /* MyBetaDist (:6-38): what every training script configures */
       QD_DIST_SQUASHED_GAUSSIAN = 1  /* MySquashedGaussian (:41-119): mean | log_std, sigmoid squashing to [0, 1] */ };
typedef struct qd_policy qd_policy;
typedef struct qd_policy_op {
  int32_t kind, in_buf, in_off, in_dim, out_buf, out_off, out_dim, act;
  int32_t flags, reserved0;     /* QD_POL_VALUE_ONLY: the op feeds only the value head, skipped when value == NULL */
  int64_t w_off, b_off;         /* float offsets into weights_host */
} qd_policy_op;
/* per-env history kept between calls (models with a time window, e.g. RMA_full's adaptation module): a ring of `rows`
 * slots of `width` floats; period 1 = one slot written per step, period 2 = two interleaved banks, each written every
 * other step (strided temporal convolutions).  At an episode start every slot holds the `width` values at
 * weights_host[fill_off] (what the network computes from the zero rows RLlib pads young episodes with). */
typedef struct qd_policy_ring {
  int32_t rows, width, period, reserved0;
  int64_t fill_off;
} qd_policy_ring;
typedef struct qd_policy_desc {
  int32_t n_ops, n_bufs;
  int32_t buf_width[8];
  int32_t obs_dim, act_dim;
  int32_t logits_buf, logits_off, n_logits;
  int32_t value_buf, value_off; /* value_buf < 0: no value head */
  int32_t n_rings;              /* 0 for feed-forward policies */
  qd_policy_ring ring[4];
  int32_t aux_buf, aux_off, aux_dim; /* an intermediate slice callers may read back (qd_policy_aux), e.g. the parameter
                                        embedding z of the RMA networks (policy.model.z, rollout.py:83); aux_dim 0 = none */
  int32_t dist;                      /* QD_DIST_*: the action distribution on the logits */
} qd_policy_desc;
size_t qd_policy_packed_bytes(const qd_policy_desc* desc, const qd_policy_op* ops);
int qd_policy_create(const qd_policy_desc* desc, const qd_policy_op* ops, const float* weights_host, size_t n_weights,
                     void* packed_device, size_t packed_bytes, qd_policy** out);
int qd_policy_destroy(qd_policy* policy);
/* device bytes of the per-env history of `num_envs` envs (0 for feed-forward policies); the caller owns the buffer and
 * passes it as `state` below.  qd_policy_reset_state fills it with the episode-start values for all envs (mask NULL) or
 * for those with mask[i] != 0; envs whose prev_truncated flag is set are re-initialised inside qd_policy_act itself. */
size_t qd_policy_state_bytes(qd_policy* policy, int num_envs);
int qd_policy_reset_state(qd_policy* policy, void* state, int num_envs, const uint8_t* mask, void* stream);
/* which kernel serves this policy: 0 = the generic layer-program interpreter, > 0 = a compile-time specialisation for
 * one of the reference's networks at the training scripts' sizes (same results; selected on an exact program match;
 * QD_POLICY_GENERIC=1 in the environment forces 0) */
int qd_policy_kernel(qd_policy* policy);
/* model.forward + value_function + MyBetaDist.deterministic_sample; any of actions / logits / value may be
 * NULL; prev_actions / prev_truncated may be NULL (= zeros / no episode boundary) */
int qd_policy_forward(qd_policy* policy, int num_envs, const float* obs, const float* prev_actions,
                      const uint8_t* prev_truncated, float* actions, float* logits, float* value, void* stream);
/* one forward pass, returning only the program's auxiliary slice: aux[N, aux_dim] (policy.model.z after compute_actions,
 * rollout.py:72,83).  Feed-forward policies only (a windowed policy's history would advance). */
int qd_policy_aux(qd_policy* policy, int num_envs, const float* obs, const float* prev_actions, const uint8_t* prev_truncated,
                  float* aux, void* stream);
/* the same forward pass with the action taken as RLlib takes it from MyBetaDist (distributions.py:6-38):
 * explore == 0: deterministic_sample (the Beta mean); explore != 0: a Beta(alpha, beta) draw (TorchBeta.sample; Philox4x32-10
 * stream keyed by `seed`, one stream per (env, counter, action dimension): pass the step number as `counter`);
 * logp[N] (nullable) = MyBetaDist.logp(action) = sum over dimensions of log Beta(clamp(a, 0.01, 0.99); alpha, beta) -- the
 * action_logp PPO stores next to the sample.  logits / value nullable as above. */
int qd_policy_act(qd_policy* policy, int num_envs, const float* obs, const float* prev_actions, const uint8_t* prev_truncated,
                  int explore, uint64_t seed, uint32_t counter, void* state, float* actions, float* logp, float* logits,
                  float* value, void* stream);
/* T closed-loop steps policy -> vector_step enqueued by one call (2 launches per step, no host round trip): what a rollout
 * worker's sampling loop does (rollout.py:64-85, RLlib's sampler): obs0[N,D] is the observation the first action is computed
 * from, prev_actions0[N,4] (nullable) the action before it; step t uses counter0 + t.  Outputs obs[T,N,D], actions[T,N,4],
 * reward[T,N], truncated[T,N] and, nullable, logp[T,N], logits[T,N,n_logits], value[T,N] -- the columns of a PPO sample
 * batch.  Same results as T x (qd_policy_act, qd_step). */
int qd_rollout_policy(qd_env* env, qd_policy* policy, int T, const float* obs0, const float* prev_actions0, int explore,
                      uint64_t seed, uint32_t counter0, void* state, float* obs, float* actions, float* reward,
                      uint8_t* truncated, float* logp, float* logits, float* value, void* stream);

/* _get_obs() on the current simulator state, obs[N,D] */
int qd_observe(qd_env* env, float* obs, void* stream);
/* get_drone_states() (BaseDroneEnv.py:357-380): states[N, qd_state_dim(model)] */
int qd_drone_states(qd_env* env, float* states, void* stream);

/* Stateless evaluation of the reference's pure functions on caller-provided state vectors
 * (rows of get_drone_states()); used by the Python mirrors of rewards.py /
 * observation_wrappers.py and by the parity tests.  ns = 33 or 29. */
int qd_eval_obs(int obs_kind, int ns, const float* states, const double ref_host[4], float* obs, int n,
                void* stream);
int qd_eval_reward(int reward_kind, int ns, const float* states, const float* actions, const int32_t* num_steps,
                   const double ref_host[4], double max_distance, float* reward, int n, void* stream);
int qd_eval_truncated(int ns, const float* states, const int32_t* num_steps, const double ref_host[4],
                      double max_distance, int max_steps, uint8_t* truncated, int n, void* stream);
/* transformation.py:5-29 on n rows */
int qd_transform(int which, const float* in, float* out, int n, void* stream);

/* ---- train-batch statistics (extension of the data path: what the reference logs about the trajectories) ----
 * MyCallbacks.on_learn_on_batch (custom_logging.py:9-31) takes train_batch['obs'] and ['actions'] to the host and logs
 * np.min / np.max / np.mean / np.var of every column.  qd_column_stats computes the same four numbers per column of a
 * row-major float32 device matrix [rows, cols] (a rollout fragment's obs [T*N, D] or actions [T*N, 4]) in one streaming
 * pass: out (device, 4*cols doubles) = min[cols] | max[cols] | mean[cols] | var[cols] (population variance, ddof 0 as
 * np.var; NaNs propagate as in numpy).  Sums are carried in float64 in a fixed order: deterministic, and at least as
 * accurate as the float32 numpy calls of the reference.  cols <= 64; rows >= 1. */
size_t qd_column_stats_workspace_bytes(int cols);
int qd_column_stats(const float* x, int64_t rows, int cols, double* out, void* workspace, size_t workspace_bytes, void* stream);
/* training.py:16-22 reads RLlib's episode_reward_mean, episode_len_mean and sum(episode_reward) / sum(episode_lengths).
 * qd_episode_stats derives them from a fragment's reward [T, N] / truncated [T, N]: an episode ends at the step whose
 * truncated flag is set (its reward counts, BaseDroneEnv.py:276-284).  carry (device, 2*N doubles: return and length of
 * every env's running episode; zero it before the first fragment) links consecutive fragments.
 * out (device, 12 doubles) = episodes ended, sum of returns, sum of lengths, sum of squared returns, min / max return,
 * min / max length, mean return, mean length, sum return / sum length, population std of the returns (NaN if none ended). */
size_t qd_episode_stats_workspace_bytes(int num_envs);
int qd_episode_stats(const float* reward, const uint8_t* truncated, int T, int num_envs, double* carry, double* out, void* workspace,
                     size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
