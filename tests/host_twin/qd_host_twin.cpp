// Test-only harness: instantiates the product's templated math headers
// (mujoco-drone_amd/csrc/qd_{model,dynamics,obsrew}.h) on the HOST, in float64
// and float32, so the specialised body-frame derivation can be compared with the
// general world-frame oracle at 1e-10 without a GPU.  Built by tests/ with g++;
// never part of the product library, which has no CPU compute path.
#include "qd_dynamics.h"
#include "qd_model.h"

using namespace qd;

template <class T>
static void run_step(int load, const double* model16, double* qpos, double* qvel, double* act, const double* ctrl,
                     double h, int nstep, double* sensor) {
  Model<T> M;
  T* mp = reinterpret_cast<T*>(&M);
  for (int i = 0; i < MODEL_FLOATS; i++) mp[i] = (T)model16[i];
  State<T> s;
  s.px = qpos[0]; s.py = qpos[1]; s.pz = qpos[2];
  s.qw = qpos[3]; s.qx = qpos[4]; s.qy = qpos[5]; s.qz = qpos[6];
  s.th1 = load ? qpos[7] : 0; s.th2 = load ? qpos[8] : 0;
  s.vx = qvel[0]; s.vy = qvel[1]; s.vz = qvel[2]; s.wx = qvel[3]; s.wy = qvel[4]; s.wz = qvel[5];
  s.thd1 = load ? qvel[6] : 0; s.thd2 = load ? qvel[7] : 0;
  s.a0 = act[0]; s.a1 = act[1]; s.a2 = act[2]; s.a3 = act[3];
  T c[4];
  for (int i = 0; i < 4; i++) c[i] = (T)(ctrl[i] < 0 ? 0 : (ctrl[i] > 1 ? 1 : ctrl[i]));
  V3<T> acc = mk<T>(0, 0, 0);
  for (int k = 0; k < nstep; k++)
    acc = load ? substep<T, true>(M, s, c[0], c[1], c[2], c[3], (T)h) : substep<T, false>(M, s, c[0], c[1], c[2], c[3], (T)h);
  qpos[0] = s.px; qpos[1] = s.py; qpos[2] = s.pz; qpos[3] = s.qw; qpos[4] = s.qx; qpos[5] = s.qy; qpos[6] = s.qz;
  qvel[0] = s.vx; qvel[1] = s.vy; qvel[2] = s.vz; qvel[3] = s.wx; qvel[4] = s.wy; qvel[5] = s.wz;
  if (load) { qpos[7] = s.th1; qpos[8] = s.th2; qvel[6] = s.thd1; qvel[7] = s.thd2; }
  act[0] = s.a0; act[1] = s.a1; act[2] = s.a2; act[3] = s.a3;
  sensor[0] = acc.x; sensor[1] = acc.y; sensor[2] = acc.z;
}

extern "C" {
int twin_derive(const double raw[6], double out16[MODEL_FLOATS]) {
  bool load;
  Model<double> M = derive_model(raw, &load);
  const double* mp = reinterpret_cast<const double*>(&M);
  for (int i = 0; i < MODEL_FLOATS; i++) out16[i] = mp[i];
  return load ? 1 : 0;
}
double twin_round5(double x) { return round5(x); }
void twin_step_f64(int load, const double* model16, double* qpos, double* qvel, double* act, const double* ctrl,
                   double h, int nstep, double* sensor) {
  run_step<double>(load, model16, qpos, qvel, act, ctrl, h, nstep, sensor);
}
void twin_step_f32(int load, const double* model16, double* qpos, double* qvel, double* act, const double* ctrl,
                   double h, int nstep, double* sensor) {
  run_step<float>(load, model16, qpos, qvel, act, ctrl, h, nstep, sensor);
}
}

// ---- observation / reward twins (float64 instantiation of qd_obsrew.h) ----
#include "qd_obsrew.h"
extern "C" {
int twin_obs(int kind, const double* s, int ns, const double* ref, double* out) {
  int n = -1;
#define CALL33(K) n = observe<double, 33, K>(s, ref, out)
#define CALL29(K) n = observe<double, 29, K>(s, ref, out)
  if (ns == 33) { QD_OBS_DISPATCH(kind, CALL33) } else { QD_OBS_DISPATCH(kind, CALL29) }
  return n;
}
int twin_obs_dim(int kind, int ns) { return obs_dim(kind, ns); }
double twin_reward(int kind, const double* s, const double* a, int k, const double* ref, double max_distance) {
  return reward<double>(kind, s, a, k, ref, max_distance);
}
int twin_truncated(const double* s, const double* ref, int k, double max_distance, int max_steps) {
  return truncated<double>(s, ref, k, max_distance, max_steps) ? 1 : 0;
}
int twin_drone_state(int load, const double* qpos, const double* qvel, const double* sens, const double* act,
                     const double* ref, const double* par, double* out) {
  State<double> s;
  s.px = qpos[0]; s.py = qpos[1]; s.pz = qpos[2]; s.qw = qpos[3]; s.qx = qpos[4]; s.qy = qpos[5]; s.qz = qpos[6];
  s.th1 = load ? qpos[7] : 0; s.th2 = load ? qpos[8] : 0;
  s.vx = qvel[0]; s.vy = qvel[1]; s.vz = qvel[2]; s.wx = qvel[3]; s.wy = qvel[4]; s.wz = qvel[5];
  s.thd1 = load ? qvel[6] : 0; s.thd2 = load ? qvel[7] : 0;
  s.a0 = act[0]; s.a1 = act[1]; s.a2 = act[2]; s.a3 = act[3];
  V3<double> acc = mk<double>(sens[0], sens[1], sens[2]);
  if (load) { drone_state<double, true>(s, acc, ref, par, out); return 33; }
  drone_state<double, false>(s, acc, ref, par, out);
  return 29;
}
void twin_simple_obs(const double* qpos, double* out) {
  State<double> s;
  s.px = qpos[0]; s.py = qpos[1]; s.pz = qpos[2]; s.qw = qpos[3]; s.qx = qpos[4]; s.qy = qpos[5]; s.qz = qpos[6];
  simple_obs<double>(s, out);
}
}

// ---- fused-kernel variants: attitude matrix taken from the quaternion ----
static State<double> mk_state(const double* qpos, const double* qvel, const double* act) {
  State<double> s;
  s.px = qpos[0]; s.py = qpos[1]; s.pz = qpos[2]; s.qw = qpos[3]; s.qx = qpos[4]; s.qy = qpos[5]; s.qz = qpos[6];
  s.th1 = qpos[7]; s.th2 = qpos[8];
  s.vx = qvel[0]; s.vy = qvel[1]; s.vz = qvel[2]; s.wx = qvel[3]; s.wy = qvel[4]; s.wz = qvel[5];
  s.thd1 = qvel[6]; s.thd2 = qvel[7];
  s.a0 = act[0]; s.a1 = act[1]; s.a2 = act[2]; s.a3 = act[3];
  return s;
}
extern "C" {
int twin_obs_q(int kind, const double* qpos, const double* qvel, const double* sens, const double* act, const double* ref,
               const double* par, double* out) {
  State<double> s = mk_state(qpos, qvel, act);
  double sv[33];
  M3<double> Rq;
  drone_state<double, true>(s, mk<double>(sens[0], sens[1], sens[2]), ref, par, sv, &Rq);
  int n = -1;
#define CALLQ(K) n = observe<double, 33, K>(sv, ref, out, &Rq)
  QD_OBS_DISPATCH(kind, CALLQ)
  return n;
}
double twin_reward_q(int kind, const double* qpos, const double* qvel, const double* sens, const double* act,
                     const double* ref, const double* par, const double* a, int k, double max_distance) {
  State<double> s = mk_state(qpos, qvel, act);
  double sv[33];
  M3<double> Rq;
  drone_state<double, true>(s, mk<double>(sens[0], sens[1], sens[2]), ref, par, sv, &Rq);
  return reward<double>(kind, sv, a, k, ref, max_distance, &Rq);
}
}

// ---- qd_pid.h (the analytic PID cascade) on the host, float64 and float32 ----
#include "qd_pid.h"
template <class T>
static void run_pid(double* st13, const double* ref, const double* xyz, const double* rpy, double mass, double force,
                    double* pos_action, double* rpyz, double* ctrl, double* action) {
  PidState<T> c;
  for (int k = 0; k < 3; k++) { c.pos_i[k] = st13[k]; c.pos_prev[k] = st13[3 + k]; c.att_i[k] = st13[6 + k]; c.att_prev[k] = st13[9 + k]; }
  c.first = (uint32_t)st13[12];
  T r4[4] = {(T)ref[0], (T)ref[1], (T)ref[2], (T)ref[3]}, p[3] = {(T)xyz[0], (T)xyz[1], (T)xyz[2]}, a[3] = {(T)rpy[0], (T)rpy[1], (T)rpy[2]};
  T pa[3], rz[4], ct[4];
  pid_position(c, r4, p, pa);
  pid_tilts2rpy(pa, r4[3], rz);
  pid_attitude(c, rz, a, (T)mass, frcp((T)force), ct);
  for (int k = 0; k < 3; k++) { pos_action[k] = pa[k]; st13[k] = c.pos_i[k]; st13[3 + k] = c.pos_prev[k]; st13[6 + k] = c.att_i[k]; st13[9 + k] = c.att_prev[k]; }
  st13[12] = c.first;
  for (int k = 0; k < 4; k++) { rpyz[k] = rz[k]; ctrl[k] = ct[k]; action[k] = qclamp(ct[k] - T(0.1), T(0), T(1)); }
}
extern "C" {
void twin_pid_f64(double* st13, const double* ref, const double* xyz, const double* rpy, double mass, double force,
                  double* pos_action, double* rpyz, double* ctrl, double* action) {
  run_pid<double>(st13, ref, xyz, rpy, mass, force, pos_action, rpyz, ctrl, action);
}
void twin_pid_f32(double* st13, const double* ref, const double* xyz, const double* rpy, double mass, double force,
                  double* pos_action, double* rpyz, double* ctrl, double* action) {
  run_pid<float>(st13, ref, xyz, rpy, mass, force, pos_action, rpyz, ctrl, action);
}
}

// ---- floor contact twin (float64 instantiation of qd_contact.h), single-body model ----
#include "qd_contact.h"
extern "C" {
// contacts of the drone's geoms with the floor: out[4 * i + {0,1,2,3}] = x, y, z, dist; returns the count
int twin_floor_contacts(double arm_len, const double* qpos, double* out) {
  double q[4] = {qpos[3], qpos[4], qpos[5], qpos[6]};
  const double n = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] * n, x = q[1] * n, y = q[2] * n, z = q[3] * n;
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                       2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
  ContactSet cs;
  contact_generate(cs, arm_len, qpos, R);
  for (int i = 0; i < cs.n; i++) { out[4 * i] = cs.x[i]; out[4 * i + 1] = cs.y[i]; out[4 * i + 2] = cs.z[i]; out[4 * i + 3] = cs.r[i]; }
  return cs.n;
}
// qacc (6) of the single-body model including the floor's reaction; returns the number of contacts
int twin_forward_floor(const double* model16, double arm_len, const double* qpos, const double* qvel, const double* act, double h,
                       double* qacc, double* force_z) {
  Model<double> M;
  double* mp = reinterpret_cast<double*>(&M);
  for (int i = 0; i < MODEL_FLOATS; i++) mp[i] = model16[i];
  State<double> s;
  s.px = qpos[0]; s.py = qpos[1]; s.pz = qpos[2]; s.qw = qpos[3]; s.qx = qpos[4]; s.qy = qpos[5]; s.qz = qpos[6];
  s.th1 = s.th2 = s.thd1 = s.thd2 = 0;
  s.vx = qvel[0]; s.vy = qvel[1]; s.vz = qvel[2]; s.wx = qvel[3]; s.wy = qvel[4]; s.wz = qvel[5];
  s.a0 = act[0]; s.a1 = act[1]; s.a2 = act[2]; s.a3 = act[3];
  Accel<double> ex, im;
  V3<double> acc;
  forward<double, false>(M, s, h, &ex, &im, &acc);
  const int n = floor_contact<double>(M, s, arm_len, h, ex.lin, ex.ang, force_z);
  qacc[0] = ex.lin.x; qacc[1] = ex.lin.y; qacc[2] = ex.lin.z; qacc[3] = ex.ang.x; qacc[4] = ex.ang.y; qacc[5] = ex.ang.z;
  return n;
}
// qacc (8) of the load model including the floor's reaction, explicit (qacc) and damping-implicit (qimp); returns the contacts
int twin_forward_floor_tree(const double* model16, const double* raw, const double* qpos, const double* qvel, const double* act, double h,
                            double* qacc, double* qimp, double* force_z) {
  Model<double> M;
  double* mp = reinterpret_cast<double*>(&M);
  for (int i = 0; i < MODEL_FLOATS; i++) mp[i] = model16[i];
  State<double> s;
  s.px = qpos[0]; s.py = qpos[1]; s.pz = qpos[2]; s.qw = qpos[3]; s.qx = qpos[4]; s.qy = qpos[5]; s.qz = qpos[6];
  s.th1 = qpos[7]; s.th2 = qpos[8];
  s.vx = qvel[0]; s.vy = qvel[1]; s.vz = qvel[2]; s.wx = qvel[3]; s.wy = qvel[4]; s.wz = qvel[5]; s.thd1 = qvel[6]; s.thd2 = qvel[7];
  s.a0 = act[0]; s.a1 = act[1]; s.a2 = act[2]; s.a3 = act[3];
  Accel<double> ex, im;
  V3<double> acc;
  forward<double, true>(M, s, h, &ex, &im, &acc);
  const int n = floor_contact_tree<double>(M, s, raw[1], raw[4], raw[5], h, ex, im, force_z);
  const double a[8] = {ex.lin.x, ex.lin.y, ex.lin.z, ex.ang.x, ex.ang.y, ex.ang.z, ex.thdd1, ex.thdd2};
  const double b[8] = {im.lin.x, im.lin.y, im.lin.z, im.ang.x, im.ang.y, im.ang.z, im.thdd1, im.thdd2};
  for (int i = 0; i < 8; i++) { qacc[i] = a[i]; qimp[i] = b[i]; }
  return n;
}
// the 8 x 8 mass matrix the contact path assembles from the three bodies' COM Jacobians (row-major)
void twin_tree_mass_matrix(const double* model16, const double* qpos, double* out64) {
  Model<double> M;
  double* mp = reinterpret_cast<double*>(&M);
  for (int i = 0; i < MODEL_FLOATS; i++) mp[i] = model16[i];
  TreePose P;
  tree_pose(qpos, qpos + 3, qpos[7], qpos[8], P);
  tree_mass_matrix(M, P, out64);
}
}

// ---- forward(): the monolithic composition the step kernels use vs the pieces the cooperative kernel uses ----
template <class T>
static void fwd_pair(const double* model16, const double* qpos, const double* qvel, const double* act, double h, double* out22) {
  Model<T> M;
  T* mp = reinterpret_cast<T*>(&M);
  for (int i = 0; i < MODEL_FLOATS; i++) mp[i] = (T)model16[i];
  State<T> s;
  s.px = qpos[0]; s.py = qpos[1]; s.pz = qpos[2]; s.qw = qpos[3]; s.qx = qpos[4]; s.qy = qpos[5]; s.qz = qpos[6];
  s.th1 = qpos[7]; s.th2 = qpos[8];
  s.vx = qvel[0]; s.vy = qvel[1]; s.vz = qvel[2]; s.wx = qvel[3]; s.wy = qvel[4]; s.wz = qvel[5];
  s.thd1 = qvel[6]; s.thd2 = qvel[7];
  s.a0 = act[0]; s.a1 = act[1]; s.a2 = act[2]; s.a3 = act[3];
  for (int which = 0; which < 2; which++) {
    Accel<T> ex, im;
    V3<T> acc;
    if (which == 0) forward<T, true>(M, s, (T)h, &ex, &im, &acc);
    else forward_pieces<T>(M, s, (T)h, &ex, &im, &acc);
    double* o = out22 + 19 * which;
    o[0] = ex.lin.x; o[1] = ex.lin.y; o[2] = ex.lin.z; o[3] = ex.ang.x; o[4] = ex.ang.y; o[5] = ex.ang.z; o[6] = ex.thdd1; o[7] = ex.thdd2;
    o[8] = im.lin.x; o[9] = im.lin.y; o[10] = im.lin.z; o[11] = im.ang.x; o[12] = im.ang.y; o[13] = im.ang.z; o[14] = im.thdd1; o[15] = im.thdd2;
    o[16] = acc.x; o[17] = acc.y; o[18] = acc.z;
  }
}
extern "C" {
void twin_forward_pair_f64(const double* model16, const double* qpos, const double* qvel, const double* act, double h, double* out38) {
  fwd_pair<double>(model16, qpos, qvel, act, h, out38);
}
void twin_forward_pair_f32(const double* model16, const double* qpos, const double* qvel, const double* act, double h, double* out38) {
  fwd_pair<float>(model16, qpos, qvel, act, h, out38);
}
}

// ---- the latency-bound kernel's arrangement of the same equations (lat_consts / mass_inverse / solve_inv5, split applied wrench) ----
template <class T>
static void fwd_lat(const double* model16, const double* qpos, const double* qvel, const double* act, double h, double* out8) {
  Model<T> M;
  T* mp = reinterpret_cast<T*>(&M);
  for (int i = 0; i < MODEL_FLOATS; i++) mp[i] = (T)model16[i];
  State<T> s;
  s.px = qpos[0]; s.py = qpos[1]; s.pz = qpos[2]; s.qw = qpos[3]; s.qx = qpos[4]; s.qy = qpos[5]; s.qz = qpos[6];
  s.th1 = qpos[7]; s.th2 = qpos[8];
  s.vx = qvel[0]; s.vy = qvel[1]; s.vz = qvel[2]; s.wx = qvel[3]; s.wy = qvel[4]; s.wz = qvel[5];
  s.thd1 = qvel[6]; s.thd2 = qvel[7];
  s.a0 = act[0]; s.a1 = act[1]; s.a2 = act[2]; s.a3 = act[3];
  Accel<T> im, ex;
  V3<T> acc;
  forward_lat<T>(M, s, (T)h, &im, &ex, &acc);
  out8[0] = im.lin.x; out8[1] = im.lin.y; out8[2] = im.lin.z; out8[3] = im.ang.x; out8[4] = im.ang.y; out8[5] = im.ang.z;
  out8[6] = im.thdd1; out8[7] = im.thdd2;
  // (the caller's buffer has 19 slots: implicit 8, explicit 8, accelerometer 3 -- forward()'s order is explicit, implicit, sensor)
  out8[8] = ex.lin.x; out8[9] = ex.lin.y; out8[10] = ex.lin.z; out8[11] = ex.ang.x; out8[12] = ex.ang.y; out8[13] = ex.ang.z;
  out8[14] = ex.thdd1; out8[15] = ex.thdd2; out8[16] = acc.x; out8[17] = acc.y; out8[18] = acc.z;
}
extern "C" {
void twin_forward_lat_f64(const double* model16, const double* qpos, const double* qvel, const double* act, double h, double* out8) {
  fwd_lat<double>(model16, qpos, qvel, act, h, out8);
}
void twin_forward_lat_f32(const double* model16, const double* qpos, const double* qvel, const double* act, double h, double* out8) {
  fwd_lat<float>(model16, qpos, qvel, act, h, out8);
}
}

// ---- accelerometer as an affine function of the activations (reset pool) vs forward() at those activations ----
extern "C" {
void twin_sensor_affine_f64(const double* model16, const double* qpos, const double* qvel, const double* act, double h, double* out6) {
  Model<double> M;
  double* mp = reinterpret_cast<double*>(&M);
  for (int i = 0; i < MODEL_FLOATS; i++) mp[i] = model16[i];
  State<double> s;
  s.px = qpos[0]; s.py = qpos[1]; s.pz = qpos[2]; s.qw = qpos[3]; s.qx = qpos[4]; s.qy = qpos[5]; s.qz = qpos[6];
  s.th1 = qpos[7]; s.th2 = qpos[8];
  s.vx = qvel[0]; s.vy = qvel[1]; s.vz = qvel[2]; s.wx = qvel[3]; s.wy = qvel[4]; s.wz = qvel[5];
  s.thd1 = qvel[6]; s.thd2 = qvel[7];
  s.a0 = act[0]; s.a1 = act[1]; s.a2 = act[2]; s.a3 = act[3];
  V3<double> c0, col[4];
  sensor_affine<double>(M, s, h, &c0, col);
  out6[0] = c0.x + act[0] * col[0].x + act[1] * col[1].x + act[2] * col[2].x + act[3] * col[3].x;
  out6[1] = c0.y + act[0] * col[0].y + act[1] * col[1].y + act[2] * col[2].y + act[3] * col[3].y;
  out6[2] = c0.z + act[0] * col[0].z + act[1] * col[1].z + act[2] * col[2].z + act[3] * col[3].z;
  Accel<double> ex, im;
  V3<double> acc;
  forward<double, true>(M, s, h, &ex, &im, &acc);
  out6[3] = acc.x; out6[4] = acc.y; out6[5] = acc.z;
}
}
// the same reading from the latency arrangement's affine form (sensor_affine_lat), float64 and float32
template <class T>
static void affine_lat(const double* model16, const double* qpos, const double* qvel, const double* act, double h, double* out3) {
  Model<T> M;
  T* mp = reinterpret_cast<T*>(&M);
  for (int i = 0; i < MODEL_FLOATS; i++) mp[i] = (T)model16[i];
  State<T> s;
  s.px = (T)qpos[0]; s.py = (T)qpos[1]; s.pz = (T)qpos[2]; s.qw = (T)qpos[3]; s.qx = (T)qpos[4]; s.qy = (T)qpos[5]; s.qz = (T)qpos[6];
  s.th1 = (T)qpos[7]; s.th2 = (T)qpos[8];
  s.vx = (T)qvel[0]; s.vy = (T)qvel[1]; s.vz = (T)qvel[2]; s.wx = (T)qvel[3]; s.wy = (T)qvel[4]; s.wz = (T)qvel[5];
  s.thd1 = (T)qvel[6]; s.thd2 = (T)qvel[7];
  s.a0 = (T)act[0]; s.a1 = (T)act[1]; s.a2 = (T)act[2]; s.a3 = (T)act[3];
  V3<T> c0, col[4];
  sensor_affine_lat<T>(M, lat_consts(M, (T)h), s, &c0, col);
  out3[0] = c0.x + s.a0 * col[0].x + s.a1 * col[1].x + s.a2 * col[2].x + s.a3 * col[3].x;
  out3[1] = c0.y + s.a0 * col[0].y + s.a1 * col[1].y + s.a2 * col[2].y + s.a3 * col[3].y;
  out3[2] = c0.z + s.a0 * col[0].z + s.a1 * col[1].z + s.a2 * col[2].z + s.a3 * col[3].z;
}
extern "C" {
void twin_sensor_affine_lat_f64(const double* model16, const double* qpos, const double* qvel, const double* act, double h, double* out3) {
  affine_lat<double>(model16, qpos, qvel, act, h, out3);
}
void twin_sensor_affine_lat_f32(const double* model16, const double* qpos, const double* qvel, const double* act, double h, double* out3) {
  affine_lat<float>(model16, qpos, qvel, act, h, out3);
}
}
