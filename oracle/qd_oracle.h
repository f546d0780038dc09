/*
 * qd_oracle.h -- CPU float64 restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it, and there only as the checker / the reported CPU
 * baseline.  The product (mujoco-drone_amd/) never links or imports it.
 *
 * What it restates (reference file:line, relative to the reference repo):
 *   - model construction      environments/env_gen.py:7-133  (+ MuJoCo's MJCF
 *                             compile step: inertia-from-geoms, 5-significant-
 *                             digit XML rounding at env_gen.py:129)
 *   - physics step            environments/mujoco_vecenv.py:404-413, i.e.
 *                             mujoco.mj_step (third-party PyPI `mujoco`,
 *                             version NOT pinned by the reference; absent from
 *                             this container).  Restated from MuJoCo's
 *                             published algorithm: forward dynamics of the
 *                             free-joint + 2-hinge tree, hinge damping,
 *                             inertia-box fluid model, first-order filter
 *                             actuators, semi-implicit Euler with implicit
 *                             joint damping, accelerometer sensor.
 *   - state extraction        environments/BaseDroneEnv.py:357-380
 *   - attitude math           environments/transformation.py:5-29
 *   - observations            environments/observation_wrappers.py:7-528,
 *                             environments/SimpleDrone.py:81-99
 *   - rewards / truncation    environments/rewards.py:5-368,
 *                             environments/BaseDroneEnv.py:12-16
 *   - reset sampling          environments/BaseDroneEnv.py:218-257
 *   - parameter randomisation environments/BaseDroneEnv.py:180-216
 *
 * Pinning status:
 *   - attitude math, observations, rewards, truncation, the reset-sampling
 *     transform and the parameter transform are PINNED by golden vectors
 *     generated from the importable reference modules (tests/golden/).
 *   - the physics step is "PARITY UNPINNED": MuJoCo is not installable here and
 *     the reference has no test or fixture that pins any mj_step result.  The
 *     dynamics are written as a general world-frame projected Newton-Euler
 *     (Kane) formulation, deliberately different from the body-frame
 *     specialised derivation in the HIP kernels, so that oracle-vs-kernel
 *     agreement cross-checks two independent derivations, and they are checked
 *     against physical invariants (tests/test_oracle_physics.py).
 */
#ifndef QD_ORACLE_H
#define QD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums: numbering must match include/qd.h ---------------------------- */
enum { ORC_MODEL_NOLOAD = 0, ORC_MODEL_LOAD = 1 };

typedef struct OrcModel {
  int    load;            /* pendulum present (pl > 0 and wm > 0)            */
  /* values as they reach MuJoCo (after %.5g rounding) */
  double gravity;         /* 9.81                                            */
  double density, viscosity, damping;
  /* body 0: attachment frame + core_body (welded)                          */
  double m0;              /* total mass of core geoms                        */
  double c0[3];           /* COM in body frame                               */
  double I0full[6];       /* xx,yy,zz,xy,xz,yz about COM, body axes          */
  double I0[3];           /* principal inertias (MuJoCo order: descending)   */
  double R0i[9];          /* inertial-frame axes in body frame (columns)     */
  double box0[3];         /* inertia-box dims, in inertial-frame axis order  */
  double rotor[4][3];     /* motor sites, body frame                         */
  double gearF;           /* thrust gear  (N per unit activation)            */
  double gearT[4];        /* yaw-torque gear, signed                         */
  double tau;             /* actuator filter time constant                   */
  double sense[3];        /* accelerometer site                              */
  /* body 1: link (hinge x)                                                  */
  double anchor[3];       /* link/pendulum joint anchor in body-0 frame      */
  double m1, I1;          /* sphere: isotropic                               */
  double box1;            /* isotropic box dim                               */
  /* body 2: pendulum (hinge y)                                              */
  double m2, lc;          /* mass, COM distance below the anchor             */
  double I2[3];           /* Ixx,Iyy,Izz about COM in pendulum frame         */
  double box2[3];
  double raw[6];          /* the six parameters the model was built from (floor contact: geom sizes) */
  double invweight[3][2]; /* MuJoCo body_invweight0 of core / link / pendulum: translational, rotational     */
} OrcModel;

/* %.5g round trip (env_gen.py:129 to_xml_string(precision=5)) */
double orc_round5g(double x);

/* env_gen.make_drone + MuJoCo compile. raw = mass, arm_len, motor_force,
 * motor_tau, pendulum_len, weight_mass (BaseDroneEnv.py:208-214). */
void orc_build_model(const double raw[6], OrcModel *out);

/* mj_forward for one drone: qacc (MuJoCo generalized coordinates: world-frame
 * linear, body-frame angular, hinge x, hinge y), act_dot and the accelerometer
 * reading.  nq = 7 (+2), nv = 6 (+2). */
void orc_forward(const OrcModel *m, const double *qpos, const double *qvel,
                 const double act[4], const double ctrl[4],
                 double *qacc, double act_dot[4], double sensor[3]);

/* mj_step x nstep (Euler, implicit joint damping).  ctrl is clamped to [0,1].
 * sensor = accelerometer of the LAST substep's forward pass. */
void orc_step(const OrcModel *m, double h, int nstep, double *qpos, double *qvel,
              double act[4], const double ctrl[4], double sensor[3]);

/* total mechanical energy and momenta (for invariant tests) */
void orc_energy_momentum(const OrcModel *m, const double *qpos, const double *qvel,
                         double *kinetic, double *potential, double lin_mom[3],
                         double ang_mom_com[3]);
/* joint-space inertia matrix (nv x nv, row-major) and bias forces */
void orc_mass_matrix(const OrcModel *m, const double *qpos, double *M);

/* transformation.py */
void orc_quat2rpy(const double q[4], double rpy[3]);
void orc_rpy2quat(const double rpy[3], double q[4]);
void orc_quat2dcm(const double q[4], double R[9]);
void orc_dcm2quat(const double R[9], double q[4]);
void orc_pendrp2quat(const double rp[2], double q[4]);

/* BaseDroneEnv.get_drone_states: 33 (load) / 29 (no load) vector */
int orc_drone_state(int load, const double *qpos, const double *qvel, const double sensor[3],
                    const double act[4], const double ref[4], const double raw[6], double *out);

/* observation variants; returns D (or -1 for the variant that raises in the
 * reference).  ns = 33 or 29. */
int orc_obs(int kind, const double *s, int ns, const double ref[4], double *out);
int orc_obs_dim(int kind, int ns);
/* SimpleDrone._get_obs for one drone: qpos[7] -> 6 */
void orc_simple_obs(const double qpos[7], double out[6]);

double orc_reward(int kind, const double *s, int ns, const double action[4], long num_steps,
                  const double ref[4], double max_distance);
int orc_truncated(const double *s, const double ref[4], long num_steps, double max_distance,
                  long max_steps);

/* Philox4x32-10 */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* sample_state: the deterministic transform from raw draws
 * (z[15] standard normals, u[2] uniforms in [0,1)) to qpos/qvel, in the
 * reference's draw order (BaseDroneEnv.py:222-242). */
typedef struct OrcSampleCfg {
  int    load, random_start;
  double start_pos[4];
  double max_pos_offset;
  double angle_var[2], vel_var[3], ang_vel_var[3], pend_rp_var[2], pend_vel_var[2];
} OrcSampleCfg;
void orc_sample_state_from_draws(const OrcSampleCfg *c, const double z[15], const double u[2],
                                 double *qpos, double *qvel);
/* the draws the device takes: Philox keyed by seed, counter (env, episode) */
void orc_sample_draws_philox(uint64_t seed, uint32_t env, uint32_t episode, float z[15], float u[2]);
/* generate_drone_params with Philox doubles: counter (env, regen) */
void orc_gen_params_philox(uint64_t seed, uint32_t env, uint32_t regen, const double center[6],
                           const double width[6], double difficulty, int random_params, int load,
                           double raw[6]);

/* batched CPU baseline: N independent drones, one full env step each
 * (ctrl map, nstep substeps, state extraction, obs, reward, truncation).
 * Layouts are AoS [N][..]; threads > 1 uses OpenMP if compiled with it. */
typedef struct OrcBatchCfg {
  int    n, load, obs_kind, reward_kind, frame_skip, ctrl_map;
  long   max_steps;
  double h, max_distance;
  double ref[4];
} OrcBatchCfg;
void orc_batch_step(const OrcBatchCfg *c, const OrcModel *models, const double *raw,
                    double *qpos, double *qvel, double *act, double *sensor, long *num_steps,
                    const double *actions, double *obs, double *reward, unsigned char *trunc,
                    int threads);

/* ---- contact of the drone's geoms (env_gen.py:41-72) with the floor plane z = 0 (env_gen.py:97) in the manner
 * of MuJoCo's soft-constraint contact model.  PARITY UNPINNED like the rest of the physics, and more so: the contact
 * generation rules (plane-box: corners below the box centre, at most 4; plane-cylinder: deepest rim point, far cap, two points
 * at 120 degrees; plane-sphere) and the constraint constants (solref 0.02 / 1, solimp 0.9 / 0.95 / 0.001 / 0.5 / 2, pyramidal
 * cone with mu = 1, diagApprox from body_invweight0, R of a pyramid edge = 2 mu^2 (1 - d)/d (1 + mu^2) invweight) are
 * restated from MuJoCo's documentation and published source from memory; nothing in the reference pins them.  The solver is
 * a projected Gauss-Seidel on the dual (the product uses a Newton method on the primal: two routes to the same minimiser). */
typedef struct OrcContact {
  double pos[3];  /* world position (midway between the surfaces, as MuJoCo places it) */
  double dist;    /* signed distance, < 0 penetrating */
  int    body;    /* 0 core, 1 link, 2 pendulum */
} OrcContact;
#define ORC_MAX_CONTACTS 64
int orc_floor_contacts(const OrcModel *m, const double *qpos, OrcContact *out);
/* orc_step with the floor: qacc = argmin of MuJoCo's convex contact problem; returns the number of contacts of the last substep */
/* mj_forward with the floor: qacc including the contact reaction (h enters through solref's 2 h floor on the time constant) */
int orc_forward_floor(const OrcModel *m, const double *qpos, const double *qvel, const double act[4], double h, double *qacc,
                      double *contact_force_z);
int orc_step_floor(const OrcModel *m, double h, int nstep, double *qpos, double *qvel, double act[4],
                   const double ctrl[4], double sensor[3], double *contact_force_z);

/* ---- the analytic cascaded PID used as a closed-loop action source
 * (models/Analytic/PositionController.py:6-34, AttitudeController.py:7-55,
 * driven as attitude_test.py:36-47).  One OrcPid per drone = the per-drone column
 * of the reference's controller objects.  Pinned by tests/golden pid_* vectors. */
typedef struct OrcPid {
  double pos_i[3], pos_prev[3];   /* PositionController.error_i / error_prev  */
  double att_i[3], att_prev[3];   /* AttittudeController.error_i / error_prev */
  int    pos_first, att_first;    /* first_step flags                         */
} OrcPid;
void orc_pid_reset(OrcPid *c);
void orc_pid_position(OrcPid *c, const double ref[3], const double xyz[3], double out[3]);
void orc_pid_tilts2rpy(const double pos_action[3], double heading_ref, double rpyz[4]);
void orc_pid_attitude(OrcPid *c, const double rpyz[4], const double rpy[3], double mass, double motor_force,
                      double ctrl[4]);
/* attitude_test.py:38-47: state -> env action (= clip(ctrl - 0.1, 0, 1)) */
void orc_pid_action(OrcPid *c, const double ref[4], const double xyz[3], const double rpy[3], double mass,
                    double motor_force, double action[4]);

/* ---- waypoint generators of evaluation.py:135-152, sample k of t = arange(0, T, dt).
 * mode 1 circle: p = {f, r, h, -}: (r cos(2 pi f t), r sin(2 pi f t), h, 0)
 * mode 2 step  : p = {step_time}:  t < step_time ? start : end
 * mode 3 ramp  : p = {start_time, duration}: t < start_time ? start : start + (t-start_time)/(duration-start_time) (end-start) */
void orc_trajectory_point(int mode, const double p[4], const double start[4], const double end[4], double dt, long k,
                          double out[4]);

#ifdef __cplusplus
}
#endif
#endif
