// qd_pid.h -- the analytic cascaded PID as an on-device action source.
//
// Restates what models/Analytic/PositionController.py:6-34 and AttitudeController.py:7-55 compute per drone
// when driven as attitude_test.py:36-47 does: position PID -> tilt / z-acceleration set-points -> desired
// attitude (roll, pitch, yaw, thrust acceleration) -> attitude PD -> motor mixer -> env action.  One drone
// per lane; the controller memory of a drone (integrators, previous errors, first-step flags) is 13 values.
#pragma once

#include "qd_math.h"

namespace qd {

constexpr uint32_t PID_POS_FIRST = 1u, PID_ATT_FIRST = 2u;

template <class T>
struct PidState {
  T pos_i[3], pos_prev[3];  // PositionController.error_i / error_prev
  T att_i[3], att_prev[3];  // AttittudeController.error_i / error_prev
  uint32_t first;           // PID_POS_FIRST | PID_ATT_FIRST: derivative disabled on the first call
};

template <class T> QD_HD void pid_reset(PidState<T>& c) {
  for (int k = 0; k < 3; k++) c.pos_i[k] = c.pos_prev[k] = c.att_i[k] = c.att_prev[k] = T(0);
  c.first = PID_POS_FIRST | PID_ATT_FIRST;
}

// PositionController.compute_control (PositionController.py:19-34): world-frame position error -> (tilt_x, tilt_y, z_acc)
template <class T> QD_HD void pid_position(PidState<T>& c, const T ref[3], const T xyz[3], T out[3]) {
  const T P[3] = {T(0.4), T(0.4), T(0.6)}, I[3] = {T(0), T(0), T(0.01)}, D[3] = {T(0.15), T(0.15), T(0.2)};
  const T dt = T(0.02), inv_dt = T(50);
  const bool first = (c.first & PID_POS_FIRST) != 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const T e = qclamp(ref[k] - xyz[k], T(-2), T(2));
    const T prev = first ? e : c.pos_prev[k];
    const T ed = (e - prev) * inv_dt;
    c.pos_prev[k] = e;
    c.pos_i[k] = qclamp(c.pos_i[k] + dt * e, T(-1), T(1));
    const T o = P[k] * e + I[k] * c.pos_i[k] + D[k] * ed;
    const T lim = k < 2 ? T(0.5) : T(2);
    out[k] = qclamp(o, -lim, lim);
  }
  c.first &= ~PID_POS_FIRST;
}

// AttittudeController.tilts2rpy (AttitudeController.py:24-39).  The reference builds Rd = [y x z, y, z] with
// y = z x heading left un-normalised and hands it to scipy's Rotation.from_matrix, which orthogonalises it
// (polar factor); for this matrix that is the same as normalising y, so Q = [y^ x z, y^, z] and the 'ZYX' Euler
// angles are read off Q directly: roll = atan2(Q21, Q22), pitch = asin(-Q20), yaw = atan2(Q10, Q00).
template <class T> QD_HD void pid_tilts2rpy(const T pa[3], T heading, T rpyz[4]) {
  T sx, cx, sy, cy, sh, ch;
  qsincos(pa[0], &sx, &cx);
  qsincos(pa[1], &sy, &cy);
  qsincos(heading, &sh, &ch);
  const T tx = sx * frcp(cx), ty = sy * frcp(cy);     // |tilt| <= 0.5 rad: cos > 0.87
  const T n2 = tx * tx + ty * ty + T(1);
  const T rn = frsq(n2);
  const V3<T> z = mk<T>(tx * rn, ty * rn, rn);
  V3<T> y = cross(z, mk<T>(ch, sh, T(0)));
  y = frsq(dot(y, y)) * y;
  const V3<T> x = cross(y, z);
  rpyz[0] = qatan2(y.z, z.z);
  rpyz[1] = qasin(qclamp(-x.z, T(-1), T(1)));
  rpyz[2] = qatan2(x.y, x.x);
  rpyz[3] = n2 * rn * qabs(pa[2] + T(9.81));          // |thrust_vec * z_acc| = sqrt(n2) |z_acc|
}

// AttittudeController.compute_control (AttitudeController.py:41-55): attitude PD + mixer -> motor commands in [0,1]
template <class T>
QD_HD void pid_attitude(PidState<T>& c, const T rpyz[4], const T rpy[3], T mass, T inv_motor_force, T ctrl[4]) {
  const T P[3] = {T(2), T(2), T(0.1)}, D[3] = {T(0.2), T(0.2), T(0)};  // I = 0: error_i is kept (and clipped) but unused
  const T dt = T(0.02), inv_dt = T(50);
  const bool first = (c.first & PID_ATT_FIRST) != 0;
  T a[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const T e = rpyz[k] - rpy[k];  // no angle wrapping in the reference
    const T prev = first ? e : c.att_prev[k];
    const T ed = (e - prev) * inv_dt;
    c.att_prev[k] = e;
    c.att_i[k] = qclamp(c.att_i[k] + dt * e, T(-1), T(1));
    a[k] = P[k] * e + D[k] * ed;
  }
  c.first &= ~PID_ATT_FIRST;
  const T base = T(0.25) * rpyz[3] * mass;
  ctrl[0] = qclamp((-a[0] - a[1] + a[2] + base) * inv_motor_force, T(0), T(1));
  ctrl[1] = qclamp(( a[0] - a[1] - a[2] + base) * inv_motor_force, T(0), T(1));
  ctrl[2] = qclamp(( a[0] + a[1] + a[2] + base) * inv_motor_force, T(0), T(1));
  ctrl[3] = qclamp((-a[0] + a[1] - a[2] + base) * inv_motor_force, T(0), T(1));
}

// state -> env action as attitude_test.py:38-47 (the env's affine control map undoes the -0.1 offset approximately)
template <class T>
QD_HD void pid_action(PidState<T>& c, const T ref[4], const T xyz[3], const T rpy[3], T mass, T motor_force, T action[4]) {
  T pa[3], rpyz[4];
  pid_position(c, ref, xyz, pa);
  pid_tilts2rpy(pa, ref[3], rpyz);
  pid_attitude(c, rpyz, rpy, mass, frcp(motor_force), action);
#pragma unroll
  for (int k = 0; k < 4; k++) action[k] = qclamp(action[k] - T(0.1), T(0), T(1));
}

// masses handed to the controller (attitude_test.py:26-29): body mass + load mass + 0.2 kg per metre of rod
// (par = mass, arm_len, motor_force, motor_tau, pendulum_len, weight_mass; the last two are 0 without a load)
template <class T> QD_HD T pid_mass(const T par[6]) { return par[0] + par[5] + T(0.2) * par[4]; }
template <class T> QD_HD T pid_motor_force(const T par[6]) { return par[2]; }

}  // namespace qd
