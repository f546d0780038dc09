// qd_obsrew.h -- per-env state vector, observation variants, rewards, truncation.
//
// Replaces the reference's three per-drone Python loops:
//   BaseDroneEnv.get_drone_states        environments/BaseDroneEnv.py:357-380
//   14 observation wrappers              environments/observation_wrappers.py:7-528
//   17 reward functions + truncation     environments/rewards.py:5-368, BaseDroneEnv.py:12-16
//   SimpleDrone._get_obs / step          environments/SimpleDrone.py:54-61,81-99
// The reference rebuilds the attitude matrix from (roll,pitch,yaw) with scipy on
// every use; here the matrix comes straight from the quaternion (same rotation).
// Reference quirks that change numbers are reproduced on purpose and marked QUIRK.
#pragma once
#include "qd_dynamics.h"
#include "qd_math.h"

namespace qd {

enum ObsKind {
  OBS_RAW = 0, OBS_GLOBAL_RPY, OBS_LOCAL_PRY, OBS_FULLSTATE, OBS_FULLSTATE_ZVEC, OBS_PRY_ACC, OBS_PRY_PARAMS,
  OBS_PRY_ACC_PARAMS, OBS_RPY_PARAMS, OBS_RPY_FAKEPARAMS, OBS_LOCAL_RPY, OBS_PRY_ACC_NOPEND,
  OBS_PRY_ACC_PARAMS_NOPEND, OBS_RM_PARAMS, OBS_ZVEC, OBS_SIMPLE, OBS_KIND_COUNT
};
enum RewardKind {
  REW_DEFAULT = 0, REW_DISTANCE, REW_DISTANCE_ENERGY, REW_PEND_ANGLE, REW_PEND_ANGLE2, REW_PEND_ANGLE3, REW_PEND_EN,
  REW_PEND_EN2, REW_PEND_EN3, REW_PEND_EN4, REW_DISTANCE_TIME_ENERGY, REW_REWARD_1, REW_PEND_DIST,
  REW_PEND_DIST_HEADING, REW_REWARD_2, REW_REWARD_2_PENERGY, REW_REWARD_3, REW_SIMPLE, REW_KIND_COUNT
};

constexpr int QD_MAX_OBS = 33;

// host/Python-side dimension of each observation variant; ns = 33 or 29
QD_HD int obs_dim(int kind, int ns) {
  const int np = ns - 27;
  switch (kind) {
    case OBS_RAW: return ns;
    case OBS_GLOBAL_RPY: case OBS_LOCAL_PRY: case OBS_LOCAL_RPY: return 16;
    case OBS_FULLSTATE: return 23;
    case OBS_FULLSTATE_ZVEC: return 24;  // QUIRK C-10: declared 23, emits 24
    case OBS_PRY_ACC: return 19;
    case OBS_PRY_PARAMS: case OBS_RPY_PARAMS: return 16 + np;
    case OBS_RPY_FAKEPARAMS: return 22;
    case OBS_PRY_ACC_PARAMS: return 19 + np;
    case OBS_PRY_ACC_NOPEND: return 15;
    case OBS_RM_PARAMS: return 22 + np;
    case OBS_ZVEC: return 17;
    case OBS_SIMPLE: return 6;
    default: return -1;  // OBS_PRY_ACC_PARAMS_NOPEND raises NameError in the reference
  }
}

// quaternion (unit) -> roll, pitch, yaw of the intrinsic ZYX decomposition (transformation.py:16-18)
template <class T>
QD_HD void quat2rpy(T w, T x, T y, T z, T* roll, T* pitch, T* yaw) {
  *roll = qatan2(T(2) * (w * x + y * z), T(1) - T(2) * (x * x + y * y));
  *pitch = qasin(qclamp(T(2) * (w * y - z * x), T(-1), T(1)));
  *yaw = qatan2(T(2) * (w * z + x * y), T(1) - T(2) * (y * y + z * z));
}

// state vector from the simulator state (BaseDroneEnv.py:365-379)
template <class T, bool LOAD>
QD_HD void drone_state(const State<T>& s, V3<T> acc, const T ref[4], const T par[6], T* o /* 33 or 29 */,
                       M3<T>* Rout = nullptr) {
  const T qn = frsq(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);
  T r, p, y;
  quat2rpy(s.qw * qn, s.qx * qn, s.qy * qn, s.qz * qn, &r, &p, &y);
  if (Rout) *Rout = quat2mat(s.qw * qn, s.qx * qn, s.qy * qn, s.qz * qn);
  o[0] = s.px; o[1] = s.py; o[2] = s.pz; o[3] = r; o[4] = p; o[5] = y;
  o[6] = s.vx; o[7] = s.vy; o[8] = s.vz; o[9] = s.wx; o[10] = s.wy; o[11] = s.wz;
  if (LOAD) {
    o[12] = s.th1; o[13] = s.th2; o[14] = s.thd1; o[15] = s.thd2;
    o[16] = acc.x; o[17] = acc.y; o[18] = acc.z; o[19] = s.a0; o[20] = s.a1; o[21] = s.a2; o[22] = s.a3;
    o[23] = ref[0]; o[24] = ref[1]; o[25] = ref[2]; o[26] = ref[3];
    o[27] = par[0]; o[28] = par[1]; o[29] = par[2]; o[30] = par[3]; o[31] = par[4]; o[32] = par[5];
  } else {
    o[12] = acc.x; o[13] = acc.y; o[14] = acc.z; o[15] = s.a0; o[16] = s.a1; o[17] = s.a2; o[18] = s.a3;
    o[19] = ref[0]; o[20] = ref[1]; o[21] = ref[2]; o[22] = ref[3];
    o[23] = par[0]; o[24] = par[1]; o[25] = par[2]; o[26] = par[3]; o[27] = par[4]; o[28] = par[5];
  }
}

// R = Rz(yaw) Ry(pitch) Rx(roll) (what mujoco_quat2DCM(mujoco_rpy2quat(rpy)) yields)
template <class T>
QD_HD M3<T> rpy2mat(T roll, T pitch, T yaw) {
  T sr, cr, sp, cp, sy, cy;
  qsincos(roll, &sr, &cr); qsincos(pitch, &sp, &cp); qsincos(yaw, &sy, &cy);
  M3<T> R;
  R.m00 = cy * cp; R.m01 = cy * sp * sr - sy * cr; R.m02 = cy * sp * cr + sy * sr;
  R.m10 = sy * cp; R.m11 = sy * sp * sr + cy * cr; R.m12 = sy * sp * cr - cy * sr;
  R.m20 = -sp;     R.m21 = cp * sr;                R.m22 = cp * cr;
  return R;
}

// observation from a state vector `s` of length NS (33 or 29).  NS and the variant are
// compile-time so that every index below is static (a runtime-indexed output array would
// be placed in scratch memory); kernels dispatch with QD_OBS_DISPATCH.  `ref` is
// env.reference (the wrappers use self.reference, not the copy inside the state).
// Returns the number of values written.
// `Rq`: optional attitude matrix already computed from the quaternion (fused kernels); without
// it the matrix is rebuilt from (roll, pitch, yaw) like the reference does.
template <class T, int NS, int kind>
QD_HD int observe(const T* s, const T ref[4], T* o, const M3<T>* Rq = nullptr) {
  constexpr int NP = NS - 27;  // QUIRK C-7: `params = state[27:]` has 6 entries with the load, 2 without
  const T pi = T(3.14159265358979323846);
  if (kind == OBS_RAW) {
#pragma unroll
    for (int i = 0; i < NS; i++) o[i] = s[i];
    return NS;
  }
  const T roll = s[3], pitch = s[4], yaw = s[5];
  const T hd = npmod(ref[3] - yaw + pi, T(2) * pi) - pi;
  const V3<T> eg = mk<T>(ref[0] - s[0], ref[1] - s[1], ref[2] - s[2]);
  if (kind == OBS_GLOBAL_RPY) {
    o[0] = eg.x; o[1] = eg.y; o[2] = eg.z; o[3] = roll; o[4] = pitch; o[5] = hd;
#pragma unroll
    for (int i = 0; i < 10; i++) o[6 + i] = s[6 + i];
    return 16;
  }
  const M3<T> R = Rq ? *Rq : rpy2mat(roll, pitch, yaw);
  const V3<T> el = mulT(R, eg), vl = mulT(R, mk<T>(s[6], s[7], s[8]));
  int n = 0;
  o[n++] = el.x; o[n++] = el.y; o[n++] = el.z;
  const bool pry = (kind == OBS_LOCAL_PRY || kind == OBS_FULLSTATE || kind == OBS_PRY_ACC || kind == OBS_PRY_PARAMS ||
                    kind == OBS_PRY_ACC_PARAMS || kind == OBS_PRY_ACC_NOPEND);
  const bool zv = (kind == OBS_FULLSTATE_ZVEC || kind == OBS_ZVEC);
  if (kind == OBS_RM_PARAMS) {
    M3<T> Rm;  // DCM([roll, pitch, -hd]); -hd = yaw - ref_yaw (mod 2 pi), so Rm = Rz(-ref_yaw) R
    if (Rq) {
      T sy, cy;
      qsincos(ref[3], &sy, &cy);
      Rm.m00 = cy * R.m00 + sy * R.m10; Rm.m01 = cy * R.m01 + sy * R.m11; Rm.m02 = cy * R.m02 + sy * R.m12;
      Rm.m10 = cy * R.m10 - sy * R.m00; Rm.m11 = cy * R.m11 - sy * R.m01; Rm.m12 = cy * R.m12 - sy * R.m02;
      Rm.m20 = R.m20; Rm.m21 = R.m21; Rm.m22 = R.m22;
    } else {
      Rm = rpy2mat(roll, pitch, -hd);
    }
    // flattened transpose
    o[n++] = Rm.m00; o[n++] = Rm.m10; o[n++] = Rm.m20; o[n++] = Rm.m01; o[n++] = Rm.m11; o[n++] = Rm.m21;
    o[n++] = Rm.m02; o[n++] = Rm.m12; o[n++] = Rm.m22;
  } else if (zv) {
    // third column of DCM([roll, pitch, 0]) = (sp cr, -sr, cp cr); from R: R20 = -sp, R21 = cp sr, R22 = cp cr
    if (Rq) {
      const T icp = frsq(qmax(R.m21 * R.m21 + R.m22 * R.m22, T(1e-30)));
      o[n++] = -R.m20 * R.m22 * icp; o[n++] = -R.m21 * icp; o[n++] = R.m22;
    } else {
      const M3<T> Rz = rpy2mat(roll, pitch, T(0));
      o[n++] = Rz.m02; o[n++] = Rz.m12; o[n++] = Rz.m22;
    }
    o[n++] = hd;
  } else {
    o[n++] = pry ? pitch : roll; o[n++] = pry ? roll : pitch; o[n++] = hd;
  }
  o[n++] = vl.x; o[n++] = vl.y; o[n++] = vl.z; o[n++] = s[9]; o[n++] = s[10]; o[n++] = s[11];
  const T prp0 = s[12], prp1 = s[13], pw0 = s[14], pw1 = s[15];
  switch (kind) {
    case OBS_LOCAL_PRY: o[n++] = prp1; o[n++] = prp0; o[n++] = pw0; o[n++] = pw1; break;
    case OBS_FULLSTATE: case OBS_FULLSTATE_ZVEC:
      o[n++] = s[16]; o[n++] = s[17]; o[n++] = s[18]; o[n++] = s[19]; o[n++] = s[20]; o[n++] = s[21]; o[n++] = s[22];
      o[n++] = prp1; o[n++] = prp0; o[n++] = pw0; o[n++] = pw1; break;
    case OBS_PRY_ACC: o[n++] = s[16]; o[n++] = s[17]; o[n++] = s[18]; o[n++] = prp1; o[n++] = prp0; o[n++] = pw0; o[n++] = pw1; break;
    case OBS_PRY_PARAMS: o[n++] = prp1; o[n++] = prp0; o[n++] = pw0; o[n++] = pw1; break;
    case OBS_PRY_ACC_PARAMS: o[n++] = prp1; o[n++] = prp0; o[n++] = s[16]; o[n++] = s[17]; o[n++] = s[18]; o[n++] = pw0; o[n++] = pw1; break;
    case OBS_PRY_ACC_NOPEND: o[n++] = s[16]; o[n++] = s[17]; o[n++] = s[18]; break;
    default: o[n++] = prp0; o[n++] = prp1; o[n++] = pw0; o[n++] = pw1; break;  // RPY_PARAMS, FAKEPARAMS, LOCAL_RPY, RM_PARAMS, ZVEC
  }
  if (kind == OBS_RPY_FAKEPARAMS) {
    o[n++] = T(1); o[n++] = T(0.17); o[n++] = T(7); o[n++] = T(0.01); o[n++] = T(1.2); o[n++] = T(0.3);
  } else if (kind == OBS_PRY_PARAMS || kind == OBS_PRY_ACC_PARAMS || kind == OBS_RPY_PARAMS || kind == OBS_RM_PARAMS) {
#pragma unroll
    for (int i = 0; i < NP; i++) o[n++] = s[27 + i];
  }
  return n;
}

// SimpleDrone._get_obs (SimpleDrone.py:94-98).  QUIRK C-9: scipy is handed MuJoCo's
// (w,x,y,z) as if it were (x,y,z,w) and asked for extrinsic 'zyx' angles.
template <class T>
QD_HD void simple_obs(const State<T>& s, T* o) {
  const T qn = frsq(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);
  const M3<T> R = quat2mat(s.qz * qn, s.qw * qn, s.qx * qn, s.qy * qn);
  o[0] = s.px; o[1] = s.py; o[2] = s.pz;
  o[3] = qatan2(-R.m01, R.m00);
  o[4] = qasin(qclamp(R.m02, T(-1), T(1)));
  o[5] = qatan2(-R.m12, R.m22);
}

// ------------------------------------------------------------------ rewards
template <class T>
QD_HD T reward(int kind, const T* s, const T a[4], int k, const T ref[4], T max_distance, const M3<T>* Rq = nullptr) {
  const T pi = T(3.14159265358979323846);
  const V3<T> dv = mk<T>(s[0] - ref[0], s[1] - ref[1], s[2] - ref[2]);
  const T d2 = dot(dv, dv);
  const T d = qsqrt(d2);
  if (kind == REW_DEFAULT) return T(3) - d;
  if (kind == REW_SIMPLE) return T(0.1) - d;
  // QUIRK: |yaw - ref_yaw| is taken BEFORE the wrap to [-pi, pi)
  const T hw = npmod(qabs(s[5] - ref[3]) + pi, T(2) * pi) - pi;
  const T h1 = qabs(hw), h2 = hw * hw;
  const T u2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
  const T kf = T(k);
  switch (kind) {
    case REW_DISTANCE: return T(5) - d - T(0.1) * h1;
    case REW_DISTANCE_ENERGY: return T(3.5) - d2 - T(0.1) * h1 - T(0.2) * u2;
    case REW_PEND_ANGLE: return T(3.5) - d2 - T(0.2) * h2 - T(0.2) * u2 - T(0.2) * (s[12] * s[12] + s[13] * s[13]);
    case REW_PEND_ANGLE2:
      return T(3.5) - d2 - T(0.5) * h2 - T(0.4) * u2 - T(0.2) * (s[12] * s[12] + s[13] * s[13]) -
             T(0.1) * (s[9] * s[9] + s[10] * s[10] + s[11] * s[11]);
    case REW_PEND_ANGLE3: {
      const T pd = s[12] * s[12] + s[13] * s[13], pv = s[14] * s[14] + s[15] * s[15];
      const T ad = s[3] * s[3] + s[4] * s[4], rs = s[9] * s[9] + s[10] * s[10] + s[11] * s[11];
      return T(3.5) - d2 - T(0.5) * h2 - T(0.4) * u2 - (T(0.1) * pd + T(0.2) * pv - T(0.3) * ad - T(0.4) * rs) / (T(1) + T(100) * d2);
    }
    case REW_DISTANCE_TIME_ENERGY: {
      const T too_far = d2 > max_distance * max_distance ? T(1) : T(0);
      return -(T(1) + T(k / 50)) * d2 - T(500) * too_far - h1 - T(0.02) * u2;
    }
    case REW_REWARD_1: {
      const T close = d2 < T(0.2) ? T(1) : T(0), too_far = d2 > max_distance * max_distance - T(3) ? T(1) : T(0);
      return (T(7) + T(20) * close - T(3) * d2 * (T(1) + kf / T(150)) - T(10) * too_far - T(0.3) * (s[3] * s[3] + s[4] * s[4]) -
              T(0.7) * h2 - T(0.3) * u2 - T(0.3) * (s[6] * s[6] + s[7] * s[7] + s[8] * s[8]) -
              T(0.5) * (s[14] * s[14] + s[15] * s[15])) / T(10);
    }
    default: break;
  }
  // the remaining rewards need the attitude matrix and the tether geometry
  const M3<T> Rd = Rq ? *Rq : rpy2mat(s[3], s[4], s[5]);
  T s1, c1, s2, c2;
  qsincos(s[12], &s1, &c1);
  qsincos(s[13], &s2, &c2);
  const V3<T> vel = mk<T>(s[6], s[7], s[8]);
  if (kind >= REW_PEND_EN && kind <= REW_PEND_EN4) {
    const T L = s[31];
    // Rp = Rx(a) Ry(b): tip = Rp (0,0,-L) = -L (s2, -s1 c2, c1 c2)
    const V3<T> tip = (-L) * mk<T>(s2, -s1 * c2, c1 * c2);
    const V3<T> om = mk<T>(s[9], s[10], s[11]);
    // (Rx ox Ry + Rx Ry oy) end, with ox = w1 [x]x, oy = w2 [y]x  (rewards.py:91-103)
    const T w1 = s[14], w2 = s[15];
    const V3<T> rel = L * mk<T>(-w2 * c2, w1 * c1 * c2 - w2 * s1 * s2, w1 * s1 * c2 + w2 * c1 * s2);
    const V3<T> c = mul(Rd, cross(om, tip) + rel);
    // QUIRK: state[6:9] (shape (3,)) + column (3,1) broadcasts to 3x3; the "energy" sums all nine squares
    const T sv = vel.x + vel.y + vel.z, sc = c.x + c.y + c.z;
    const T E = T(3) * dot(vel, vel) + T(2) * sv * sc + T(3) * dot(c, c);
    const T ph = mul(Rd, tip).z;
    const T ad = qsqrt(s[3] * s[3] + s[4] * s[4] + s[5] * s[5]);
    if (kind == REW_PEND_EN) return T(3.5) - d2 - T(0.5) * h2 - T(0.4) * u2 - T(0.2) * E;
    const T thr = (kind == REW_PEND_EN4) ? T(0.6) : T(0.5);
    T ce = T(0);
    { T m0 = qmax(a[0] - thr, T(0)), m1 = qmax(a[1] - thr, T(0)), m2 = qmax(a[2] - thr, T(0)), m3 = qmax(a[3] - thr, T(0));
      ce = m0 * m0 + m1 * m1 + m2 * m2 + m3 * m3; }
    if (kind == REW_PEND_EN2) {
      T r = T(3.5) - T(2) * d - T(0.6) * h2 - T(0.6) * ce;
      if (d < T(0.15)) r = r + T(3) - T(0.2) * E - T(0.2) * ad;
      return r;
    }
    const T tot = T(0.5) * E + T(9.81) * ph;
    if (kind == REW_PEND_EN3) return T(7) - d - T(0.4) * h2 - T(0.1) * ce - T(0.1) * tot - T(0.05) * ad;
    return T(5) - d - T(0.6) * h2 - T(0.1) * ce - (T(0.2) * tot + T(0.05) * ad) / (T(0.5) + d);
  }
  // QUIRK C-8: these use the ZYX matrix Ry(prp1) Rx(prp0) for the tether, and
  // reward_pendulum_dist uses params[5] (load mass) as the tether length
  const T L = (kind == REW_PEND_DIST) ? s[32] : s[31];
  // Rp' (0,0,-L) with Rp' = Ry(p) Rx(r): third column of Ry Rx = (sp cr, -sr, cp cr)  (r = prp0, p = prp1)
  const V3<T> rl = (-L) * mk<T>(s2 * c1, -s1, c2 * c1);
  const V3<T> tipw = mul(Rd, rl);
  const V3<T> e = mk<T>(s[0] + tipw.x - ref[0], s[1] + tipw.y - ref[1], s[2] + tipw.z - ref[2]);
  const T dp2 = dot(e, e);
  if (kind == REW_PEND_DIST) return -dp2;
  if (kind == REW_PEND_DIST_HEADING) return T(3) - dp2 - T(0.1) * h1;
  if (kind == REW_REWARD_2) return T(4) - dp2 - T(0.001) * kf * dp2 - T(0.1) * h1 - T(0.05) * u2;
  const V3<T> vg = vel + mul(Rd, cross(mk<T>(s[14], s[15], T(0)), rl));
  const T Ep = dot(vg, vg);
  if (kind == REW_REWARD_2_PENERGY)
    return T(4) - dp2 - T(0.2) * h1 - T(0.006) * kf * (dp2 + T(0.2) * h1) - T(0.05) * u2 - T(0.1) * Ep;
  T cm;
  { T m0 = qmin(a[0] - T(0.5), T(0)), m1 = qmin(a[1] - T(0.5), T(0)), m2 = qmin(a[2] - T(0.5), T(0)), m3 = qmin(a[3] - T(0.5), T(0));
    cm = m0 * m0 + m1 * m1 + m2 * m2 + m3 * m3; }
  return T(4) - d2 - T(0.2) * h1 - T(0.006) * kf * (d2 + T(0.2) * h1 + T(0.01) * Ep) - T(0.1) * cm - T(0.1) * Ep;
}

// runtime -> compile-time dispatch over the observation variant (wave-uniform branch)
#define QD_OBS_DISPATCH(kind, CALL)                                                      \
  switch (kind) {                                                                        \
    case 0: CALL(0); break;   case 1: CALL(1); break;   case 2: CALL(2); break;          \
    case 3: CALL(3); break;   case 4: CALL(4); break;   case 5: CALL(5); break;          \
    case 6: CALL(6); break;   case 7: CALL(7); break;   case 8: CALL(8); break;          \
    case 9: CALL(9); break;   case 10: CALL(10); break; case 11: CALL(11); break;        \
    case 13: CALL(13); break; case 14: CALL(14); break; default: break;                  \
  }

// default_termination_fcn (BaseDroneEnv.py:12-16); returned as `truncated`
// A non-finite position also truncates: in the reference a diverged state never reaches this test because MuJoCo's
// mj_checkPos/mj_checkAcc reset the data first (which then lies far from the reference and truncates one step later).
template <class T>
QD_HD bool truncated(const T* s, const T ref[4], int num_steps, T max_distance, int max_steps) {
  const V3<T> dv = mk<T>(s[0] - ref[0], s[1] - ref[1], s[2] - ref[2]);
  return !(qsqrt(dot(dv, dv)) <= max_distance) || num_steps >= max_steps;
}

}  // namespace qd
