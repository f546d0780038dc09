// qd_dynamics.h -- rigid-body dynamics of one quadrotor (+ two-hinge tether and
// load), re-derived for one-env-per-lane execution.
//
// Replaces mujoco.mj_step as the reference calls it
// (environments/mujoco_vecenv.py:404-413) for the model of env_gen.py:7-73:
//   bodies   0 core (free joint), 1 link sphere (hinge x), 2 rod+load (hinge y)
//   passive  hinge damping, inertia-box fluid drag on every body
//   actuators first-order filtered rotor thrust + yaw reaction torque
//   integrator semi-implicit Euler, joint damping implicit ((M + h D) a = f)
//   sensor   accelerometer at the core (pre-integration acceleration)
//
// Derivation (NOT MuJoCo's world-frame CRB/RNE): everything is expressed in the
// core body frame F0.  Generalised accelerations A = (a0, alpha, thdd1, thdd2)
// with a0 the classical acceleration of the body origin in F0 components
// (world acceleration = R a0), alpha the body-frame angular acceleration.
// Gravity enters as the fictitious origin acceleration g~ = R^T (0,0,9.81).
// The tether body is axially symmetric, I2 = It*1 + (Ia-It) d d^T with d the
// unit tether direction, so all of its terms are written with d directly.
// Mass matrix structure exploited:
//   * linear block is mt*1            -> eliminated analytically (COM reduction)
//   * hinge 2x2 block is diagonal, hinge axes are orthogonal to each other and
//     axis 2 is orthogonal to d
//   * what is left is one symmetric 3x3 factorisation (LDL^T) shared by the
//     explicit solve (sensor) and the damping-implicit solve (integration), then
//     two 2x2 solves.
//
// Precision: state, trigonometry, fluid forces and integration are float32 (T).  The
// velocity-product terms, the inertia assembly about the system COM (sums of
// non-negative terms, no cancellation) and the solves of the load model run in HP:
// float64 on the device.  The load hangs 1.2 m below a 0.01 kg m^2 airframe, so the 3x3
// inertia has a condition number of ~60 and float32 algebra there costs a factor ~15 in
// 200-step trajectory divergence (measured: 5e-4 -> 3e-5 worst case over 128 envs with
// random rotor commands, against the float64 oracle); CDNA4 issues f64 FMAs at half the
// packed-f32 rate, i.e. at the rate of the scalar f32 FMAs this per-lane code uses anyway.
#pragma once
#include "qd_math.h"
#include "qd_model.h"

namespace qd {

template <class T>
struct State {
  T px, py, pz;          // world position of the body origin      qpos[0:3]
  T qw, qx, qy, qz;      // attitude quaternion (w,x,y,z)           qpos[3:7]
  T th1, th2;            // hinge x, hinge y                        qpos[7:9]
  T vx, vy, vz;          // world-frame linear velocity             qvel[0:3]
  T wx, wy, wz;          // body-frame angular velocity             qvel[3:6]
  T thd1, thd2;          // hinge rates                             qvel[6:8]
  T a0, a1, a2, a3;      // actuator activations                    act[0:4]
};

template <class T>
struct Accel {
  V3<T> lin;   // world-frame linear acceleration of the origin (qacc[0:3])
  V3<T> ang;   // body-frame angular acceleration               (qacc[3:6])
  T thdd1, thdd2;
};

// the float64 type used for the algebra core of the load model
template <class T> struct HighPrec { using type = double; };

// inertia-box fluid wrench from precomputed coefficients (qd_model.h): local angular
// velocity w and local COM velocity v, both in the body's inertial-frame axes
template <class T>
QD_HD void fluid(T klin, T kang, T qlx, T qly, T qlz, T qax, T qay, T qaz, V3<T> w, V3<T> v, V3<T>* f, V3<T>* tq) {
  f->x = -(klin + qlx * qabs(v.x)) * v.x;
  f->y = -(klin + qly * qabs(v.y)) * v.y;
  f->z = -(klin + qlz * qabs(v.z)) * v.z;
  tq->x = -(kang + qax * qabs(w.x)) * w.x;
  tq->y = -(kang + qay * qabs(w.y)) * w.y;
  tq->z = -(kang + qaz * qabs(w.z)) * w.z;
}

template <class A, class B> QD_HD V3<A> cvt(V3<B> v) { return mk<A>(A(v.x), A(v.y), A(v.z)); }

// solve with the LDL^T factor of the 3x3 rotational block (F: any struct with d0, d1, d2 = reciprocal pivots, l10, l20, l21)
template <class F, class HP>
QD_HD V3<HP> ldl_solve(const F& f, V3<HP> b) {
  const HP y0 = b.x, y1 = b.y - f.l10 * y0;
  const HP y2 = b.z - f.l20 * y0 - f.l21 * y1;
  const HP z2 = y2 * f.d2;
  const HP z1 = y1 * f.d1 - f.l21 * z2;
  const HP z0 = y0 * f.d0 - f.l10 * z1 - f.l20 * z2;
  return mk<HP>(z0, z1, z2);
}

// ---- the forward dynamics of the load model, in four pieces -----------------------------------------------
// forward() below is their composition in one lane.  They are separate because the cooperative step kernel
// (k_step_coop, qd_kernels.hip) runs them in three wavefronts of a workgroup at once: the applied wrench (thrust +
// fluid drag, float32), the inertial wrench (gravity + velocity products, HP) and the mass-matrix factorisation (HP)
// do not depend on each other, only the final solve needs all three.

// attitude-dependent quantities shared by the pieces
template <class T>
struct Att {
  M3<T> R;      // body -> world
  V3<T> w;      // body-frame angular velocity
  V3<T> vb;     // origin velocity in body axes
  V3<T> gt;     // fictitious origin acceleration R^T (0,0,g)
  V3<T> u;      // w x (w x zhat): velocity-product acceleration per unit height on the body z axis
};
template <class T>
QD_HD Att<T> attitude(const State<T>& s) {
  Att<T> a;
  // MuJoCo normalises the stored quaternion before use
  const T qn = frsq(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);
  a.R = quat2mat(s.qw * qn, s.qx * qn, s.qy * qn, s.qz * qn);
  a.w = mk<T>(s.wx, s.wy, s.wz);
  a.vb = mulT(a.R, mk<T>(s.vx, s.vy, s.vz));
  const T g = T(Const::gravity);
  a.gt = mk<T>(g * a.R.m20, g * a.R.m21, g * a.R.m22);
  a.u = mk<T>(a.w.x * a.w.z, a.w.y * a.w.z, -(a.w.x * a.w.x + a.w.y * a.w.y));
  return a;
}
// only the gravity direction and the angular velocity (what the inertial wrench needs of the attitude)
template <class T>
QD_HD void gravity_body(const State<T>& s, V3<T>* gt, V3<T>* w) {
  const T qn2 = frcp(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);   // = qn * qn of attitude()
  const T g2 = T(2) * T(Const::gravity) * qn2;
  *gt = mk<T>(g2 * (s.qx * s.qz - s.qw * s.qy), g2 * (s.qy * s.qz + s.qw * s.qx), T(Const::gravity) - g2 * (s.qx * s.qx + s.qy * s.qy));
  *w = mk<T>(s.wx, s.wy, s.wz);
}

// tether geometry from the two hinge angles (float32 trigonometry on the device)
template <class T>
struct Tether {
  T s1, c1, s2, c2;
  V3<T> d;     // unit vector anchor -> load (F0 axes)
  V3<T> y2;    // hinge-2 axis (F0 axes); hinge-1 axis is x
  V3<T> e_x;   // x axis of the tether frame F2 = Rx Ry (its z axis is -d)
};
template <class T>
QD_HD Tether<T> tether_geometry(T th1, T th2) {
  Tether<T> t;
  qsincos(th1, &t.s1, &t.c1);
  qsincos(th2, &t.s2, &t.c2);
  t.d = mk<T>(-t.s2, t.s1 * t.c2, -t.c1 * t.c2);
  t.y2 = mk<T>(T(0), t.c1, t.s1);
  t.e_x = mk<T>(t.c2, t.s1 * t.s2, -t.c1 * t.s2);
  return t;
}

// the same from the two sine / cosine pairs (the latency kernel computes them once per step, in the solver wave)
template <class T>
QD_HD Tether<T> tether_from_trig(T s1, T c1, T s2, T c2) {
  Tether<T> t;
  t.s1 = s1; t.c1 = c1; t.s2 = s2; t.c2 = c2;
  t.d = mk<T>(-t.s2, t.s1 * t.c2, -t.c1 * t.c2);
  t.y2 = mk<T>(T(0), t.c1, t.s1);
  t.e_x = mk<T>(t.c2, t.s1 * t.s2, -t.c1 * t.s2);
  return t;
}

// applied wrench: rotor thrust + inertia-box fluid drag on the three bodies, reduced to
//   F  total force, Tq total torque about the body origin (both F0 axes), t1 / t2 hinge torques
template <class T>
struct Applied {
  V3<T> F, Tq;
  T t1, t2;
};
template <class T>
QD_HD Applied<T> applied_wrench(const Model<T>& M, const State<T>& s, const Att<T>& at, const Tether<T>& tg) {
  const V3<T> w = at.w, vb = at.vb;
  // rotors (env_gen.py:53-64): thrust along body z at (+-rot, +-rot, 0), yaw reaction +-gearT
  const T f0 = M.gearF * s.a0, f1 = M.gearF * s.a1, f2 = M.gearF * s.a2, f3 = M.gearF * s.a3;
  const V3<T> fT = mk<T>(T(0), T(0), f0 + f1 + f2 + f3);
  const V3<T> tT = mk<T>(M.rot * (-f0 + f1 + f2 - f3), M.rot * (-f0 - f1 + f2 + f3), M.gearT * (s.a0 - s.a1 + s.a2 - s.a3));
  // fluid drag on the core, evaluated in body axes (see qd_model.h on the principal frame); COM at (0,0,c0z): v_com = vb + w x c0
  V3<T> fD0, tD0;
  fluid(M.klin0, M.kang0, M.qlx0, M.qly0, M.qlz0, M.qax0, M.qay0, M.qaz0, w, mk<T>(vb.x + w.y * M.c0z, vb.y - w.x * M.c0z, vb.z), &fD0, &tD0);
  // fluid drag on link (frame F1 = Rx(th1)) and tether (frame F2)
  const T az = T(Const::anchor_z), s1 = tg.s1, c1 = tg.c1;
  const V3<T> d = tg.d, y2 = tg.y2, e_x = tg.e_x;
  const V3<T> w1 = mk<T>(w.x + s.thd1, w.y, w.z);
  const V3<T> w2 = w1 + s.thd2 * y2;
  const V3<T> va = mk<T>(vb.x + w.y * az, vb.y - w.x * az, vb.z);  // anchor velocity
  const V3<T> vc2 = va + M.lc * cross(w2, d);                      // tether COM velocity
  const T k1 = T(LinkFluid::klin), k2 = T(LinkFluid::kang), k3 = T(LinkFluid::ql), k4 = T(LinkFluid::qa);
  // F0 -> F1 components: Rx^T v = (x, c1 y + s1 z, -s1 y + c1 z)
  V3<T> fl, tl;
  fluid(k1, k2, k3, k3, k3, k4, k4, k4, mk<T>(w1.x, c1 * w1.y + s1 * w1.z, -s1 * w1.y + c1 * w1.z),
        mk<T>(va.x, c1 * va.y + s1 * va.z, -s1 * va.y + c1 * va.z), &fl, &tl);
  const V3<T> fD1 = mk<T>(fl.x, c1 * fl.y - s1 * fl.z, s1 * fl.y + c1 * fl.z);
  const V3<T> tD1 = mk<T>(tl.x, c1 * tl.y - s1 * tl.z, s1 * tl.y + c1 * tl.z);
  fluid(M.klin2, M.kang2, M.qlt2, M.qlt2, M.qla2, M.qat2, M.qat2, M.qaa2,
        mk<T>(dot(e_x, w2), dot(y2, w2), -dot(d, w2)), mk<T>(dot(e_x, vc2), dot(y2, vc2), -dot(d, vc2)), &fl, &tl);
  const V3<T> fD2 = fl.x * e_x + fl.y * y2 - fl.z * d;
  const V3<T> tD2 = tl.x * e_x + tl.y * y2 - tl.z * d;
  // reduction: wrench on the tether about the anchor, then everything about the origin
  const V3<T> rho = M.lc * d;
  const V3<T> t2v = tD2 + cross(rho, fD2);
  const V3<T> f12 = fD1 + fD2;
  Applied<T> ap;
  ap.F = fT + fD0 + f12;
  ap.Tq = tT + tD0 + mk<T>(-M.c0z * fD0.y, M.c0z * fD0.x, T(0)) + tD1 + mk<T>(-az * f12.y, az * f12.x, T(0)) + t2v;
  ap.t1 = tD1.x + t2v.x;
  ap.t2 = dot(y2, t2v);
  return ap;
}

// inertial wrench: the same reduction of mass x acceleration / Euler terms with all generalised accelerations zero
// and origin acceleration g~ (gravity + velocity products), plus the passive hinge damping; HP arithmetic
template <class HP>
struct Inertial {
  V3<HP> F, Tq;
  HP t1, t2;
};
// CORE = false: without the core body's share (its weight and gyroscopic moment), which applied_core_link<true>() then carries in
// float32 on the applied side -- no precision is lost there that the rotor thrust beside it had kept
template <class T, class HP, bool CORE = true>
QD_HD Inertial<HP> inertial_wrench_hp(const Model<T>& M, const State<T>& s, V3<T> gt, V3<T> w, V3<HP> dh, V3<HP> y2h) {
  const HP m0 = M.m0, m1 = Const::m1, m2 = M.m2, i1 = Const::I1, It = M.I2t, lc = M.lc;
  const HP c0z = M.c0z, azh = Const::anchor_z;
  const HP dI = HP(M.I2a) - It;
  const V3<HP> wh = cvt<HP>(w), gth = cvt<HP>(gt);
  const V3<HP> uh = mk<HP>(wh.x * wh.z, wh.y * wh.z, -(wh.x * wh.x + wh.y * wh.y));
  const HP thd1 = s.thd1, thd2 = s.thd2;
  const V3<HP> rho = lc * dh;                                         // anchor -> tether COM
  const V3<HP> w1 = mk<HP>(wh.x + thd1, wh.y, wh.z);
  const V3<HP> w2 = w1 + thd2 * y2h;
  // accelerations with all generalised accelerations zero and origin acceleration g~
  const V3<HP> aa = gth + azh * uh;                                   // anchor
  const V3<HP> al1 = mk<HP>(HP(0), thd1 * wh.z, -thd1 * wh.y);        // thd1 * (w x xhat)
  const V3<HP> al2 = al1 + thd2 * cross(w1, y2h);
  const V3<HP> ac2 = aa + cross(al2, rho) + dot(w2, rho) * w2 - dot(w2, w2) * rho;
  const V3<HP> F1 = m1 * aa;
  const V3<HP> N1 = i1 * al1;
  const V3<HP> F2 = m2 * ac2;
  const V3<HP> N2 = It * al2 + (dI * dot(dh, al2)) * dh + (dI * dot(dh, w2)) * cross(w2, dh);
  const V3<HP> n2v = N2 + cross(rho, F2);
  const V3<HP> F12 = F1 + F2;
  const HP bd = Const::damping;
  Inertial<HP> in;
  if (CORE) {
    const V3<HP> F0 = m0 * (gth + c0z * uh);
    const V3<HP> N0 = mk<HP>(wh.y * wh.z * (HP(M.I0z) - HP(M.I0y)), wh.z * wh.x * (HP(M.I0x) - HP(M.I0z)),
                             wh.x * wh.y * (HP(M.I0y) - HP(M.I0x)));
    in.F = F0 + F12;
    in.Tq = N0 + mk<HP>(-c0z * F0.y, c0z * F0.x, HP(0)) + N1 + mk<HP>(-azh * F12.y, azh * F12.x, HP(0)) + n2v;
  } else {
    in.F = F12;
    in.Tq = N1 + mk<HP>(-azh * F12.y, azh * F12.x, HP(0)) + n2v;
  }
  in.t1 = N1.x + n2v.x + bd * thd1;
  in.t2 = dot(y2h, n2v) + bd * thd2;
  return in;
}
template <class T>
QD_HD Inertial<typename HighPrec<T>::type> inertial_wrench(const Model<T>& M, const State<T>& s, V3<T> gt, V3<T> w, const Tether<T>& tg) {
  using HP = typename HighPrec<T>::type;
  return inertial_wrench_hp<T, HP, true>(M, s, gt, w, cvt<HP>(tg.d), cvt<HP>(tg.y2));
}
// The hinge angles' sine / cosine pairs in HP with s^2 + c^2 = 1 to HP rounding: one Newton step of the inverse square root about 1
// (the float32 pair is off the unit circle by ~1e-7).  mass_inverse() uses identities of a UNIT tether direction (d.d = 1, d.y2 = 0);
// fed with the float32 pair as is, its J and B D^-1 B^T -- two numbers of size m2 lc^2 whose difference is the airframe's inertia --
// disagree at 1e-7 and the accelerations at 2e-4 (host twin).  The inertial wrench of the same step takes the same pairs, so that the
// load's weight has exactly no moment about the COM the solve uses.
template <class HP, class T>
QD_HD void trig_unit(T sf, T cf, HP* s, HP* c) {
  const HP s0 = sf, c0 = cf;
  const HP f = HP(1.5) - HP(0.5) * (s0 * s0 + c0 * c0);
  *s = s0 * f; *c = c0 * f;
}
template <class HP>
struct TetherHP {
  HP s1, c1, s2, c2;
  V3<HP> d, y2;
};
template <class HP, class T>
QD_HD TetherHP<HP> tether_hp(T s1f, T c1f, T s2f, T c2f) {
  TetherHP<HP> t;
  trig_unit(s1f, c1f, &t.s1, &t.c1);
  trig_unit(s2f, c2f, &t.s2, &t.c2);
  t.d = mk<HP>(-t.s2, t.s1 * t.c2, -t.c1 * t.c2);
  t.y2 = mk<HP>(HP(0), t.c1, t.s1);
  return t;
}

// mass matrix about the system COM (sums of non-negative terms), LDL^T of its 3x3 rotational block, the two hinge
// columns solved against it and the 2x2 Schur complement on the hinges; HP arithmetic
template <class HP>
struct Factor {
  HP d0, d1, d2, l10, l20, l21;   // LDL^T (d = reciprocal pivots)
  V3<HP> B1, B2, X1, X2;          // hinge columns of the mass matrix and J^-1 of them
  HP s11, s12, s22;               // Schur complement (without the implicit damping h*b on its diagonal)
  V3<HP> S, rc, p1, p2;           // first moment, COM, xhat x rho, y2 x rho
  HP imt, m2;
  // folded here so that the solve (the serial part of the step: everything waits for it) is as short as possible
  V3<HP> Sm, kp1, kp2;            // imt * S, (m2 imt) * p1, (m2 imt) * p2
  HP idet_ex, idet_im, hb;        // 1 / det of the Schur complement without / with the implicit damping hb on its diagonal
  HP ixx, ixy, ixz, iyy, iyz, izz;  // J^-1 written out: a right-hand side then costs a 3-deep product instead of a 9-deep substitution
};
// PRE: also fold the products and reciprocals the solve needs (cooperative kernel: the factorisation has slack, the solve
// does not); without it they are formed where they are used, which keeps fewer values alive (single-lane composition)
template <bool PRE = true, class T>
QD_HD Factor<typename HighPrec<T>::type> mass_factor(const Model<T>& M, const Tether<T>& tg, T h) {
  using HP = typename HighPrec<T>::type;
  Factor<HP> f;
  const HP m0 = M.m0, m1 = Const::m1, m2 = M.m2, i1 = Const::I1, It = M.I2t, lc = M.lc;
  const HP c0z = M.c0z, azh = Const::anchor_z;
  const HP dI = HP(M.I2a) - It;
  const V3<HP> dh = cvt<HP>(tg.d), y2h = cvt<HP>(tg.y2);
  const V3<HP> rho = lc * dh;
  const HP mt = m0 + m1 + m2, imt = frcp(mt);
  const V3<HP> r2 = mk<HP>(rho.x, rho.y, rho.z + azh);
  const V3<HP> S = mk<HP>(m2 * r2.x, m2 * r2.y, m0 * c0z + m1 * azh + m2 * r2.z);
  const V3<HP> rc = imt * S;
  const HP c1h = tg.c1, s1h = tg.s1, c2h = tg.c2, s2h = tg.s2;
  const V3<HP> p1 = lc * mk<HP>(HP(0), c1h * c2h, s1h * c2h);          // xhat x rho
  const V3<HP> p2 = lc * mk<HP>(-c2h, -s1h * s2h, c1h * s2h);          // y2 x rho
  const V3<HP> q2 = r2 - rc;
  const HP qx = -rc.x, qy = -rc.y, q0z = c0z - rc.z, q1z = azh - rc.z; // core and link COM offsets from the COM
  const HP m01 = m0 + m1, base = i1 + It;
  const HP zz01 = m0 * q0z * q0z + m1 * q1z * q1z, z01 = m0 * q0z + m1 * q1z;
  const HP Jxx = HP(M.I0x) + base + dI * dh.x * dh.x + m01 * qy * qy + zz01 + m2 * (q2.y * q2.y + q2.z * q2.z);
  const HP Jyy = HP(M.I0y) + base + dI * dh.y * dh.y + m01 * qx * qx + zz01 + m2 * (q2.x * q2.x + q2.z * q2.z);
  const HP Jzz = HP(M.I0z) + base + dI * dh.z * dh.z + m01 * (qx * qx + qy * qy) + m2 * (q2.x * q2.x + q2.y * q2.y);
  const HP Jxy = dI * dh.x * dh.y - m01 * qx * qy - m2 * q2.x * q2.y;
  const HP Jxz = dI * dh.x * dh.z - qx * z01 - m2 * q2.x * q2.z;
  const HP Jyz = dI * dh.y * dh.z - qy * z01 - m2 * q2.y * q2.z;
  f.B1 = mk<HP>(base, HP(0), HP(0)) + (dI * dh.x) * dh + m2 * cross(q2, p1);
  f.B2 = It * y2h + m2 * cross(q2, p2);
  const HP mu = m2 * (mt - m2) * imt;
  const HP lc2 = lc * lc;
  const HP D1 = base + dI * s2h * s2h + mu * lc2 * c2h * c2h;
  const HP D2 = It + mu * lc2;
  // LDL^T of the 3x3 block
  f.d0 = frcp(Jxx);
  f.l10 = Jxy * f.d0; f.l20 = Jxz * f.d0;
  f.d1 = frcp(Jyy - f.l10 * Jxy);
  const HP t21 = Jyz - f.l20 * Jxy;
  f.l21 = t21 * f.d1;
  f.d2 = frcp(Jzz - f.l20 * Jxz - f.l21 * t21);
  f.X1 = ldl_solve(f, f.B1);
  f.X2 = ldl_solve(f, f.B2);
  f.s11 = D1 - dot(f.B1, f.X1); f.s12 = -dot(f.B1, f.X2); f.s22 = D2 - dot(f.B2, f.X2);
  f.S = S; f.rc = rc; f.p1 = p1; f.p2 = p2; f.imt = imt; f.m2 = m2;
  f.hb = HP(h) * HP(Const::damping);
  if (PRE) {
    const HP k = m2 * imt;
    f.Sm = imt * S; f.kp1 = k * p1; f.kp2 = k * p2;
    f.idet_ex = frcp(f.s11 * f.s22 - f.s12 * f.s12);
    f.idet_im = frcp((f.s11 + f.hb) * (f.s22 + f.hb) - f.s12 * f.s12);
    // J^-1 = L^-T D^-1 L^-1 column by column (unit vectors through the factor)
    const V3<HP> cx = ldl_solve(f, mk<HP>(HP(1), HP(0), HP(0))), cy = ldl_solve(f, mk<HP>(HP(0), HP(1), HP(0)));
    f.ixx = cx.x; f.ixy = cx.y; f.ixz = cx.z; f.iyy = cy.y; f.iyz = cy.z; f.izz = f.d2;
  }
  return f;
}

// right-hand side of the reduced system from the two wrenches
template <class HP>
struct Rhs {
  V3<HP> fl, Xf;   // net force; J^-1 of the net torque about the COM
  HP q1, q2;       // hinge right-hand sides after eliminating the rotational block
};
template <bool PRE = true, class T, class HP>
QD_HD Rhs<HP> reduce_rhs(const Factor<HP>& f, const Applied<T>& ap, const Inertial<HP>& in) {
  Rhs<HP> r;
  r.fl = cvt<HP>(ap.F) - in.F;
  const V3<HP> fw = cvt<HP>(ap.Tq) - in.Tq;
  const HP ft1 = HP(ap.t1) - in.t1, ft2 = HP(ap.t2) - in.t2;
  const V3<HP> fwr = fw - cross(f.rc, r.fl);
  const HP k = f.m2 * f.imt;
  const HP g1 = PRE ? ft1 - dot(f.kp1, r.fl) : ft1 - k * dot(f.p1, r.fl);
  const HP g2 = PRE ? ft2 - dot(f.kp2, r.fl) : ft2 - k * dot(f.p2, r.fl);
  if (PRE) r.Xf = mk<HP>(f.ixx * fwr.x + f.ixy * fwr.y + f.ixz * fwr.z, f.ixy * fwr.x + f.iyy * fwr.y + f.iyz * fwr.z,
                         f.ixz * fwr.x + f.iyz * fwr.y + f.izz * fwr.z);
  else r.Xf = ldl_solve(f, fwr);
  r.q1 = g1 - dot(f.B1, r.Xf);
  r.q2 = g2 - dot(f.B2, r.Xf);
  return r;
}

// generalised accelerations: IMPLICIT = false: damping explicit (what MuJoCo stores in qacc); true: the damping-implicit
// Euler update (M + h D) a = M qacc, i.e. h * damping on the hinge diagonal.  a0 = origin acceleration in body axes.
template <bool IMPLICIT, bool PRE = true, class T, class HP>
QD_HD void finish_accel(const Factor<HP>& f, const Rhs<HP>& r, V3<HP>* a0, V3<T>* ang, T* thdd1, T* thdd2) {
  const HP S11 = IMPLICIT ? f.s11 + f.hb : f.s11, S22 = IMPLICIT ? f.s22 + f.hb : f.s22;
  const HP idet = PRE ? (IMPLICIT ? f.idet_im : f.idet_ex) : frcp(S11 * S22 - f.s12 * f.s12);
  const HP t1 = (S22 * r.q1 - f.s12 * r.q2) * idet, t2 = (S11 * r.q2 - f.s12 * r.q1) * idet;
  const V3<HP> al = r.Xf - t1 * f.X1 - t2 * f.X2;
  if (PRE) *a0 = f.imt * r.fl - cross(al, f.Sm) - t1 * f.kp1 - t2 * f.kp2;
  else *a0 = f.imt * (r.fl - cross(al, f.S) - (f.m2 * t1) * f.p1 - (f.m2 * t2) * f.p2);
  *ang = cvt<T>(al); *thdd1 = T(t1); *thdd2 = T(t2);
}

// accelerometer reading (site frame = body frame) from the damping-explicit accelerations
template <class T>
QD_HD V3<T> accelerometer(V3<T> a0e, V3<T> ang_ex, V3<T> gt, V3<T> u) {
  const T sz = T(Const::sense_z);
  return a0e + gt + mk<T>(ang_ex.y * sz, -ang_ex.x * sz, T(0)) + sz * u;
}

// ---- the same reduced system, arranged for the latency-bound persistent kernel (k_rollout_lat, qd_rollout_lat.hip) ----------
// mass_factor() above costs ~175 float64 instructions per step, most of them the assembly of the inertia about the system COM
// from its definition.  Everything in that assembly is a function of the unit tether direction d alone, with coefficients that
// depend on the env's parameters only: with r2 = lc d + az z (tether COM), S = a d + b z (first moment, a = m2 lc,
// b = m0 c0z + m1 az + m2 az) and Steiner's theorem taken about the ORIGIN and shifted to the COM once,
//     J  = diag(Kx, Ky, Kz) - 2 k3 dz 1 + k2 d d^T + k3 (d z^T + z d^T)          (Jzz = Kz + k2 dz^2: the dz terms cancel)
//     B1 = (b1c - k3 dz) x + k2 dx d              B2 = (d2c - k3 dz) y2 + k3 s1 d
//     D1 = base + dI s2^2 + mul2 c2^2             D2 = d2c            rc = kl d + rz0 z       (m2/mt) p1,2 = kl (x, y2) x d
// (k2 = dI - mu lc^2, k3 = -m2 lc (az - b/mt), mul2 = mu lc^2, mu = m2 (mt - m2) / mt: the reduced mass of the tether against
// the rest).  LatConsts holds those coefficients, computed once per launch in float64.
// The elimination order is turned round as well: the hinge block is DIAGONAL, so the hinges go first (two reciprocals, one of
// them a per-env constant) and the 3x3 that is left, J' = J - e1 B1 B1^T - e2 B2 B2^T, is inverted by its adjugate (one more
// reciprocal) instead of an LDL^T with three; the result is kept as the symmetric inverse of the whole 5x5 rotational + hinge
// system, so that the solve -- the serial half of a step -- is 5 dot products of 5 terms:
//     [alpha]   [ C    -U1   -U2 ] [fwr]
//     [thdd1] = [-U1^T  s11   s12 ] [g1 ]        C = J'^-1, U_i = C (e_i B_i), s_ij = delta_ij e_i + (e_i B_i) . U_j
//     [thdd2]   [-U2^T  s12   s22 ] [g2 ]
// Only the damping-implicit system (e_i = 1 / (D_i + h b), what the integration uses) is formed: the accelerometer reading needs
// the explicit one, which differs from it by h b on the two hinge diagonals -- explicit_weights / explicit_from_implicit below
// get its solution from this one's by a 2 x 2 correction.
// Same equations as forward(): tests/test_host_twin.py::test_latency_pieces_equal_the_monolithic_forward, 1e-11 in float64.
template <class HP>
struct LatConsts {
  HP Kx, Ky, Kz, k2, k3, b1c, d2c, base, dI, mul2, kl, rz0, imt, hb, e2;
};
template <class T>
QD_HD LatConsts<typename HighPrec<T>::type> lat_consts(const Model<T>& M, T h) {
  using HP = typename HighPrec<T>::type;
  LatConsts<HP> k;
  const HP m0 = M.m0, m1 = Const::m1, m2 = M.m2, i1 = Const::I1, It = M.I2t, lc = M.lc, c0z = M.c0z, az = Const::anchor_z;
  const HP mt = m0 + m1 + m2, imt = HP(1) / mt;
  const HP a = m2 * lc, b = m0 * c0z + m1 * az + m2 * az;
  const HP z2 = m0 * c0z * c0z + m1 * az * az;
  k.base = i1 + It;
  k.dI = HP(M.I2a) - It;
  const HP mu = m2 * (mt - m2) * imt;
  k.mul2 = mu * lc * lc;
  const HP common = k.base + m2 * (lc * lc + az * az) - (a * a + b * b) * imt;
  k.Kx = HP(M.I0x) + z2 + common;
  k.Ky = HP(M.I0y) + z2 + common;
  k.Kz = HP(M.I0z) + k.base + m2 * lc * lc - a * a * imt;
  k.k2 = k.dI - k.mul2;
  k.k3 = -m2 * lc * (az - b * imt);
  k.b1c = k.base + k.mul2;
  k.d2c = It + k.mul2;
  k.kl = a * imt;
  k.rz0 = b * imt;
  k.imt = imt;
  k.hb = HP(h) * HP(Const::damping);
  k.e2 = HP(1) / (k.d2c + k.hb);
  return k;
}
template <class HP>
struct Inv5 {
  HP cxx, cxy, cxz, cyy, cyz, czz;   // C
  V3<HP> U1, U2;
  HP s11, s12, s22;
  V3<HP> rc, kp1, kp2;               // COM; (m2 / mt) (x, y2) x rho
};
// th: tether_hp() of the float32 sine / cosine pairs (unit to HP rounding)
template <class HP>
QD_HD Inv5<HP> mass_inverse(const LatConsts<HP>& k, const TetherHP<HP>& th) {
  Inv5<HP> v;
  const HP s1 = th.s1, c1 = th.c1, s2 = th.s2, c2 = th.c2;
  const HP dx = th.d.x, dy = th.d.y, dz = th.d.z;
  const HP k2x = k.k2 * dx, k2y = k.k2 * dy, k2z = k.k2 * dz, shift = HP(-2) * k.k3 * dz;
  const HP Jxx = k.Kx + shift + k2x * dx, Jyy = k.Ky + shift + k2y * dy, Jzz = k.Kz + k2z * dz;
  const HP Jxy = k2x * dy, Jxz = k2x * dz + k.k3 * dx, Jyz = k2y * dz + k.k3 * dy;
  const HP w1 = k.b1c - k.k3 * dz, w2 = k.d2c - k.k3 * dz, k3s = k.k3 * s1;
  const V3<HP> B1 = mk<HP>(w1 + k2x * dx, k2x * dy, k2x * dz);
  const V3<HP> B2 = mk<HP>(k3s * dx, w2 * c1 + k3s * dy, w2 * s1 + k3s * dz);
  const HP D1 = k.base + k.dI * s2 * s2 + k.mul2 * c2 * c2;
  const HP e1 = frcp(D1 + k.hb), e2 = k.e2;
  const V3<HP> E1 = e1 * B1, E2 = e2 * B2;
  // J' = J - e1 B1 B1^T - e2 B2 B2^T
  const HP Pxx = Jxx - E1.x * B1.x - E2.x * B2.x, Pxy = Jxy - E1.x * B1.y - E2.x * B2.y, Pxz = Jxz - E1.x * B1.z - E2.x * B2.z;
  const HP Pyy = Jyy - E1.y * B1.y - E2.y * B2.y, Pyz = Jyz - E1.y * B1.z - E2.y * B2.z, Pzz = Jzz - E1.z * B1.z - E2.z * B2.z;
  // adjugate
  const HP a00 = Pyy * Pzz - Pyz * Pyz, a01 = Pxz * Pyz - Pxy * Pzz, a02 = Pxy * Pyz - Pxz * Pyy;
  const HP a11 = Pxx * Pzz - Pxz * Pxz, a12 = Pxy * Pxz - Pxx * Pyz, a22 = Pxx * Pyy - Pxy * Pxy;
  const HP idet = frcp(Pxx * a00 + Pxy * a01 + Pxz * a02);
  v.cxx = a00 * idet; v.cxy = a01 * idet; v.cxz = a02 * idet; v.cyy = a11 * idet; v.cyz = a12 * idet; v.czz = a22 * idet;
  v.U1 = mk<HP>(v.cxx * E1.x + v.cxy * E1.y + v.cxz * E1.z, v.cxy * E1.x + v.cyy * E1.y + v.cyz * E1.z, v.cxz * E1.x + v.cyz * E1.y + v.czz * E1.z);
  v.U2 = mk<HP>(v.cxx * E2.x + v.cxy * E2.y + v.cxz * E2.z, v.cxy * E2.x + v.cyy * E2.y + v.cyz * E2.z, v.cxz * E2.x + v.cyz * E2.y + v.czz * E2.z);
  v.s11 = e1 + dot(E1, v.U1); v.s12 = dot(E1, v.U2); v.s22 = e2 + dot(E2, v.U2);
  v.rc = mk<HP>(k.kl * dx, k.kl * dy, k.kl * dz + k.rz0);
  v.kp1 = mk<HP>(HP(0), -k.kl * dz, k.kl * dy);                                   // kl (x x d)
  v.kp2 = mk<HP>(k.kl * (c1 * dz - s1 * dy), k.kl * (s1 * dx), -k.kl * (c1 * dx));  // kl (y2 x d)
  return v;
}
// the damping-implicit generalised accelerations from the two wrenches (what reduce_rhs + finish_accel<true> give), in two stages: the
// angular and hinge accelerations (kept in HP: explicit_from_implicit() below starts from them), then the origin's
template <class HP>
struct Rot5 {
  V3<HP> fl, al;
  HP t1, t2;
};
template <class T, class HP>
QD_HD Rot5<HP> solve_inv5_rot(const Inv5<HP>& v, const Applied<T>& ap, const Inertial<HP>& in) {
  Rot5<HP> r;
  r.fl = cvt<HP>(ap.F) - in.F;
  const V3<HP> fw = cvt<HP>(ap.Tq) - in.Tq;
  const HP ft1 = HP(ap.t1) - in.t1, ft2 = HP(ap.t2) - in.t2;
  const V3<HP> fwr = fw - cross(v.rc, r.fl);
  const HP g1 = ft1 - (v.kp1.y * r.fl.y + v.kp1.z * r.fl.z), g2 = ft2 - dot(v.kp2, r.fl);
  r.al = mk<HP>(v.cxx * fwr.x + v.cxy * fwr.y + v.cxz * fwr.z - g1 * v.U1.x - g2 * v.U2.x,
                v.cxy * fwr.x + v.cyy * fwr.y + v.cyz * fwr.z - g1 * v.U1.y - g2 * v.U2.y,
                v.cxz * fwr.x + v.cyz * fwr.y + v.czz * fwr.z - g1 * v.U1.z - g2 * v.U2.z);
  r.t1 = v.s11 * g1 + v.s12 * g2 - dot(v.U1, fwr);
  r.t2 = v.s12 * g1 + v.s22 * g2 - dot(v.U2, fwr);
  return r;
}
template <class HP>
QD_HD V3<HP> solve_inv5_lin(const LatConsts<HP>& k, const Inv5<HP>& v, const Rot5<HP>& r) {
  return k.imt * r.fl - cross(r.al, v.rc) - r.t1 * v.kp1 - r.t2 * v.kp2;
}
template <class T, class HP>
QD_HD void solve_inv5(const LatConsts<HP>& k, const Inv5<HP>& v, const Applied<T>& ap, const Inertial<HP>& in, V3<HP>* a0, V3<T>* ang, T* thdd1,
                      T* thdd2) {
  const Rot5<HP> r = solve_inv5_rot(v, ap, in);
  *a0 = solve_inv5_lin(k, v, r);
  *ang = cvt<T>(r.al); *thdd1 = T(r.t1); *thdd2 = T(r.t2);
}

// The damping-EXPLICIT accelerations (what MuJoCo stores in qacc and the accelerometer reads) from the implicit solve, without a second
// inverse: the two systems differ by h b on the two hinge diagonals, A_ex = A_im - h b E E^T, so by Woodbury
//     x_ex = x_im + G E W E^T x_im,    W = (1 / (h b) - E^T G E)^-1 = (1 / (h b) - [s11 s12; s12 s22])^-1    (2 x 2),
// with G E = the inverse's two hinge columns (-U1, -U2 | s): four multiply-adds for W (t1, t2), ten for the correction.
template <class HP>
struct ExW {
  HP w11, w12, w22;
};
template <class HP>
QD_HD ExW<HP> explicit_weights(const LatConsts<HP>& k, const Inv5<HP>& v) {
  const HP ih = HP(1) / k.hb;
  const HP a = ih - v.s11, d = ih - v.s22, b = -v.s12;
  const HP idet = frcp(a * d - b * b);
  ExW<HP> w;
  w.w11 = d * idet; w.w22 = a * idet; w.w12 = -b * idet;
  return w;
}
template <class T, class HP>
QD_HD void explicit_from_implicit(const LatConsts<HP>& k, const Inv5<HP>& v, const ExW<HP>& w, V3<HP> fl, V3<HP> al, HP t1, HP t2, V3<HP>* a0ex,
                                  V3<T>* ang, T* thdd1, T* thdd2) {
  const HP d1 = w.w11 * t1 + w.w12 * t2, d2 = w.w12 * t1 + w.w22 * t2;
  const V3<HP> ale = al - d1 * v.U1 - d2 * v.U2;
  const HP t1e = t1 + v.s11 * d1 + v.s12 * d2, t2e = t2 + v.s12 * d1 + v.s22 * d2;
  *a0ex = k.imt * fl - cross(ale, v.rc) - t1e * v.kp1 - t2e * v.kp2;
  *ang = cvt<T>(ale); *thdd1 = T(t1e); *thdd2 = T(t2e);
}

// applied_wrench() in two halves, for two wavefronts: rotors + drag on core and link | drag on the tether.  Their sum is the
// applied wrench (the order of the float32 additions differs from applied_wrench(): rounding-level differences).
// CORE_INERTIAL: minus the core body's inertial wrench (weight in body axes gt = R^T (0,0,g), gyroscopic moment), in T
template <bool CORE_INERTIAL = false, class T>
QD_HD Applied<T> applied_core_link(const Model<T>& M, const State<T>& s, V3<T> w, V3<T> vb, T s1, T c1, V3<T> gt = V3<T>{T(0), T(0), T(0)}) {
  const T f0 = M.gearF * s.a0, f1 = M.gearF * s.a1, f2 = M.gearF * s.a2, f3 = M.gearF * s.a3;
  const V3<T> tT = mk<T>(M.rot * (-f0 + f1 + f2 - f3), M.rot * (-f0 - f1 + f2 + f3), M.gearT * (s.a0 - s.a1 + s.a2 - s.a3));
  V3<T> fD0, tD0;
  fluid(M.klin0, M.kang0, M.qlx0, M.qly0, M.qlz0, M.qax0, M.qay0, M.qaz0, w, mk<T>(vb.x + w.y * M.c0z, vb.y - w.x * M.c0z, vb.z), &fD0, &tD0);
  const T az = T(Const::anchor_z);
  const V3<T> w1 = mk<T>(w.x + s.thd1, w.y, w.z);
  const V3<T> va = mk<T>(vb.x + w.y * az, vb.y - w.x * az, vb.z);
  const T k1 = T(LinkFluid::klin), k2 = T(LinkFluid::kang), k3 = T(LinkFluid::ql), k4 = T(LinkFluid::qa);
  V3<T> fl, tl;
  fluid(k1, k2, k3, k3, k3, k4, k4, k4, mk<T>(w1.x, c1 * w1.y + s1 * w1.z, -s1 * w1.y + c1 * w1.z),
        mk<T>(va.x, c1 * va.y + s1 * va.z, -s1 * va.y + c1 * va.z), &fl, &tl);
  const V3<T> fD1 = mk<T>(fl.x, c1 * fl.y - s1 * fl.z, s1 * fl.y + c1 * fl.z);
  const V3<T> tD1 = mk<T>(tl.x, c1 * tl.y - s1 * tl.z, s1 * tl.y + c1 * tl.z);
  Applied<T> ap;
  ap.F = mk<T>(fD0.x + fD1.x, fD0.y + fD1.y, (f0 + f1 + f2 + f3) + fD0.z + fD1.z);
  ap.Tq = tT + tD0 + mk<T>(-M.c0z * fD0.y, M.c0z * fD0.x, T(0)) + tD1 + mk<T>(-az * fD1.y, az * fD1.x, T(0));
  ap.t1 = tD1.x;
  ap.t2 = T(0);
  if (CORE_INERTIAL) {
    const V3<T> u = mk<T>(w.x * w.z, w.y * w.z, -(w.x * w.x + w.y * w.y));
    const V3<T> F0 = M.m0 * (gt + M.c0z * u);
    const V3<T> N0 = mk<T>(w.y * w.z * (M.I0z - M.I0y), w.z * w.x * (M.I0x - M.I0z), w.x * w.y * (M.I0y - M.I0x));
    ap.F = ap.F - F0;
    ap.Tq = ap.Tq - N0 - mk<T>(-M.c0z * F0.y, M.c0z * F0.x, T(0));
  }
  return ap;
}
template <class T>
QD_HD Applied<T> applied_tether(const Model<T>& M, const State<T>& s, V3<T> w, V3<T> vb, const Tether<T>& tg) {
  const T az = T(Const::anchor_z);
  const V3<T> d = tg.d, y2 = tg.y2, e_x = tg.e_x;
  const V3<T> w1 = mk<T>(w.x + s.thd1, w.y, w.z);
  const V3<T> w2 = w1 + s.thd2 * y2;
  const V3<T> va = mk<T>(vb.x + w.y * az, vb.y - w.x * az, vb.z);
  const V3<T> vc2 = va + M.lc * cross(w2, d);
  V3<T> fl, tl;
  fluid(M.klin2, M.kang2, M.qlt2, M.qlt2, M.qla2, M.qat2, M.qat2, M.qaa2,
        mk<T>(dot(e_x, w2), dot(y2, w2), -dot(d, w2)), mk<T>(dot(e_x, vc2), dot(y2, vc2), -dot(d, vc2)), &fl, &tl);
  const V3<T> fD2 = fl.x * e_x + fl.y * y2 - fl.z * d;
  const V3<T> tD2 = tl.x * e_x + tl.y * y2 - tl.z * d;
  const V3<T> t2v = tD2 + cross(M.lc * d, fD2);
  Applied<T> ap;
  ap.F = fD2;
  ap.Tq = mk<T>(-az * fD2.y, az * fD2.x, T(0)) + t2v;
  ap.t1 = t2v.x;
  ap.t2 = dot(y2, t2v);
  return ap;
}
// body-frame quantities both halves need: attitude matrix of the normalised quaternion and the origin velocity in body axes
template <class T>
QD_HD void attitude_min(const State<T>& s, M3<T>* R, V3<T>* vb) {
  const T qn = frsq(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);
  *R = quat2mat(s.qw * qn, s.qx * qn, s.qy * qn, s.qz * qn);
  *vb = mulT(*R, mk<T>(s.vx, s.vy, s.vz));
}

// The same forward dynamics composed from the latency pieces in one lane (implicit accelerations only; host twin test).
template <class T>
QD_HD void forward_lat(const Model<T>& M, const State<T>& s, T h, Accel<T>* im, Accel<T>* ex = nullptr, V3<T>* acc = nullptr) {
  using HP = typename HighPrec<T>::type;
  M3<T> R;
  V3<T> vb;
  attitude_min(s, &R, &vb);
  const V3<T> w = mk<T>(s.wx, s.wy, s.wz);
  const Tether<T> tg = tether_geometry(s.th1, s.th2);
  const T g = T(Const::gravity);
  const Applied<T> a1 = applied_core_link<true>(M, s, w, vb, tg.s1, tg.c1, mk<T>(g * R.m20, g * R.m21, g * R.m22)), a2 = applied_tether(M, s, w, vb, tg);
  Applied<T> ap;
  ap.F = a1.F + a2.F; ap.Tq = a1.Tq + a2.Tq; ap.t1 = a1.t1 + a2.t1; ap.t2 = a1.t2 + a2.t2;
  V3<T> gt, w_;
  gravity_body(s, &gt, &w_);
  const TetherHP<HP> th = tether_hp<HP>(tg.s1, tg.c1, tg.s2, tg.c2);
  const Inertial<HP> in = inertial_wrench_hp<T, HP, false>(M, s, gt, w_, th.d, th.y2);
  const LatConsts<HP> k = lat_consts(M, h);
  const Inv5<HP> v = mass_inverse(k, th);
  V3<HP> a0;
  solve_inv5(k, v, ap, in, &a0, &im->ang, &im->thdd1, &im->thdd2);
  im->lin = mul(R, cvt<T>(a0));
  if (ex) {
    const Rot5<HP> r5 = solve_inv5_rot(v, ap, in);   // (the implicit solution once more, in HP)
    const V3<HP> fl = r5.fl, al = r5.al;
    const HP t1 = r5.t1, t2 = r5.t2;
    V3<HP> a0ex;
    explicit_from_implicit(k, v, explicit_weights(k, v), fl, al, t1, t2, &a0ex, &ex->ang, &ex->thdd1, &ex->thdd2);
    const V3<T> a0e = cvt<T>(a0ex);
    ex->lin = mul(R, a0e);
    if (acc) {
      const T g = T(Const::gravity);
      *acc = accelerometer(a0e, ex->ang, mk<T>(g * R.m20, g * R.m21, g * R.m22), mk<T>(w.x * w.z, w.y * w.z, -(w.x * w.x + w.y * w.y)));
    }
  }
}

// The accelerometer reading as an affine function of the activations at a fixed state: reading(a) = c0 + sum_i a_i col_i.
// Thrust and yaw reaction are linear in the activations and nothing else in the forward dynamics depends on them, so one
// factorisation serves the five right-hand sides.  Used by the reset pool (qd_kernels.hip): an env's activations survive a
// reset (reference quirk C-2) and are only known when it happens, but c0 and the four columns can be prepared with the
// pre-sampled state, which turns mj_forward's sensor refresh at reset time into 12 multiply-adds.
template <class T>
QD_HD void sensor_affine(const Model<T>& M, State<T> s, T h, V3<T>* c0, V3<T> col[4]) {
  using HP = typename HighPrec<T>::type;
  s.a0 = s.a1 = s.a2 = s.a3 = T(0);
  const Att<T> at = attitude(s);
  const Tether<T> tg = tether_geometry(s.th1, s.th2);
  const Applied<T> ap = applied_wrench(M, s, at, tg);
  const Inertial<HP> in = inertial_wrench(M, s, at.gt, at.w, tg);
  const Factor<HP> f = mass_factor<true>(M, tg, h);
  const T sz = T(Const::sense_z);
  {
    const Rhs<HP> r = reduce_rhs<true>(f, ap, in);
    V3<HP> a0;
    V3<T> ang;
    T d1, d2;
    finish_accel<false, true>(f, r, &a0, &ang, &d1, &d2);
    *c0 = accelerometer(cvt<T>(a0), ang, at.gt, at.u);
  }
  Inertial<HP> zero;
  zero.F = mk<HP>(HP(0), HP(0), HP(0)); zero.Tq = zero.F; zero.t1 = zero.t2 = HP(0);
  // unit activation of rotor i: thrust gearF along z at (+-rot, +-rot, 0), yaw reaction +-gearT (applied_wrench)
  const T sx[4] = {T(-1), T(1), T(1), T(-1)}, sy[4] = {T(-1), T(-1), T(1), T(1)}, sg[4] = {T(1), T(-1), T(1), T(-1)};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    Applied<T> d;
    d.F = mk<T>(T(0), T(0), M.gearF);
    d.Tq = mk<T>(M.rot * M.gearF * sx[i], M.rot * M.gearF * sy[i], M.gearT * sg[i]);
    d.t1 = d.t2 = T(0);
    const Rhs<HP> r = reduce_rhs<true>(f, d, zero);
    V3<HP> a0;
    V3<T> ang;
    T d1, d2;
    finish_accel<false, true>(f, r, &a0, &ang, &d1, &d2);
    const V3<T> a0e = cvt<T>(a0);
    col[i] = mk<T>(a0e.x + ang.y * sz, a0e.y - ang.x * sz, a0e.z);
  }
}

// sensor_affine() in the latency arrangement (the closed policy loop's reset lanes, qd_rollout_fused.hip): one mass_inverse serves
// the five right-hand sides, each solve 5 dot products + the 2 x 2 correction to the explicit system -- 30 registers of inverse
// where the LDL^T factor holds 100 (k: lat_consts(M, h), which the caller has).  Same reading: tests/test_host_twin.py, against
// forward() at random activations.
template <class T>
QD_HD void sensor_affine_lat(const Model<T>& M, const LatConsts<typename HighPrec<T>::type>& k, State<T> s, V3<T>* c0, V3<T> col[4]) {
  using HP = typename HighPrec<T>::type;
  s.a0 = s.a1 = s.a2 = s.a3 = T(0);
  M3<T> R;
  V3<T> vb;
  attitude_min(s, &R, &vb);
  const V3<T> w = mk<T>(s.wx, s.wy, s.wz);
  const Tether<T> tg = tether_geometry(s.th1, s.th2);
  const T g = T(Const::gravity), sz = T(Const::sense_z);
  const V3<T> gtR = mk<T>(g * R.m20, g * R.m21, g * R.m22);
  const Applied<T> a1 = applied_core_link<true>(M, s, w, vb, tg.s1, tg.c1, gtR), a2 = applied_tether(M, s, w, vb, tg);
  Applied<T> ap;
  ap.F = a1.F + a2.F; ap.Tq = a1.Tq + a2.Tq; ap.t1 = a1.t1 + a2.t1; ap.t2 = a1.t2 + a2.t2;
  V3<T> gt, w_;
  gravity_body(s, &gt, &w_);
  const TetherHP<HP> th = tether_hp<HP>(tg.s1, tg.c1, tg.s2, tg.c2);
  const Inertial<HP> in = inertial_wrench_hp<T, HP, false>(M, s, gt, w_, th.d, th.y2);
  const Inv5<HP> v = mass_inverse(k, th);
  const ExW<HP> xw = explicit_weights(k, v);
  V3<HP> a0ex;
  V3<T> ang;
  T d1, d2;
  {
    const Rot5<HP> r = solve_inv5_rot(v, ap, in);
    explicit_from_implicit(k, v, xw, r.fl, r.al, r.t1, r.t2, &a0ex, &ang, &d1, &d2);
    *c0 = accelerometer(cvt<T>(a0ex), ang, gtR, mk<T>(w.x * w.z, w.y * w.z, -(w.x * w.x + w.y * w.y)));
  }
  Inertial<HP> zero;
  zero.F = mk<HP>(HP(0), HP(0), HP(0)); zero.Tq = zero.F; zero.t1 = zero.t2 = HP(0);
  // unit activation of rotor i: thrust gearF along z at (+-rot, +-rot, 0), yaw reaction +-gearT (applied_core_link)
  const T sx[4] = {T(-1), T(1), T(1), T(-1)}, sy[4] = {T(-1), T(-1), T(1), T(1)}, sg[4] = {T(1), T(-1), T(1), T(-1)};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    Applied<T> d;
    d.F = mk<T>(T(0), T(0), M.gearF);
    d.Tq = mk<T>(M.rot * M.gearF * sx[i], M.rot * M.gearF * sy[i], M.gearT * sg[i]);
    d.t1 = d.t2 = T(0);
    const Rot5<HP> r = solve_inv5_rot(v, d, zero);
    explicit_from_implicit(k, v, xw, r.fl, r.al, r.t1, r.t2, &a0ex, &ang, &d1, &d2);
    const V3<T> a0e = cvt<T>(a0ex);
    col[i] = mk<T>(a0e.x + ang.y * sz, a0e.y - ang.x * sz, a0e.z);
  }
}

// The same forward dynamics as forward() below, composed from the pieces above in one lane (load model).  The step
// kernels do NOT use it: hipcc schedules the monolithic forward() with fewer live values (36 vs 52 bytes of scratch in the
// 256-thread instantiation, 12-16 % of the step time in the HBM-bound regime), so forward() stays as it was and the pieces
// serve the cooperative kernel.  tests/test_host_twin.py holds the two together: identical to 1e-12 in float64.
template <class T>
QD_HD void forward_pieces(const Model<T>& M, const State<T>& s, T h, Accel<T>* ex, Accel<T>* im, V3<T>* acc) {
  using HP = typename HighPrec<T>::type;
  const Att<T> at = attitude(s);
  const Tether<T> tg = tether_geometry(s.th1, s.th2);
  const Applied<T> ap = applied_wrench(M, s, at, tg);
  V3<T> gt, w;
  gravity_body(s, &gt, &w);
  const Inertial<HP> in = inertial_wrench(M, s, gt, w, tg);
  const Factor<HP> f = mass_factor<true>(M, tg, h);
  const Rhs<HP> r = reduce_rhs<true>(f, ap, in);
  V3<HP> a0ex, a0im;
  finish_accel<false, true>(f, r, &a0ex, &ex->ang, &ex->thdd1, &ex->thdd2);
  finish_accel<true, true>(f, r, &a0im, &im->ang, &im->thdd1, &im->thdd2);
  const V3<T> a0e = cvt<T>(a0ex);
  *acc = accelerometer(a0e, ex->ang, at.gt, at.u);
  ex->lin = mul(at.R, a0e);
  im->lin = mul(at.R, cvt<T>(a0im));
}

// forward dynamics at the current state.
//   ex  : accelerations with damping explicit (what MuJoCo stores in qacc; feeds the sensor)
//   im  : accelerations of the damping-implicit Euler update ((M + h D) a = M qacc)
//   acc : accelerometer reading (site frame = body frame)
template <class T, bool LOAD>
QD_HD void forward(const Model<T>& M, const State<T>& s, T h, Accel<T>* ex, Accel<T>* im, V3<T>* acc) {
  using HP = typename HighPrec<T>::type;
  // attitude (MuJoCo normalises the stored quaternion before use)
  const T qn = frsq(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);
  const M3<T> R = quat2mat(s.qw * qn, s.qx * qn, s.qy * qn, s.qz * qn);
  const V3<T> w = mk<T>(s.wx, s.wy, s.wz);
  const V3<T> vb = mulT(R, mk<T>(s.vx, s.vy, s.vz));  // origin velocity in body axes
  const T g = T(Const::gravity);
  const V3<T> gt = mk<T>(g * R.m20, g * R.m21, g * R.m22);

  // rotors (env_gen.py:53-64): thrust along body z at (+-rot, +-rot, 0), yaw reaction +-gearT
  const T f0 = M.gearF * s.a0, f1 = M.gearF * s.a1, f2 = M.gearF * s.a2, f3 = M.gearF * s.a3;
  const V3<T> fT = mk<T>(T(0), T(0), f0 + f1 + f2 + f3);
  const V3<T> tT = mk<T>(M.rot * (-f0 + f1 + f2 - f3), M.rot * (-f0 - f1 + f2 + f3), M.gearT * (s.a0 - s.a1 + s.a2 - s.a3));
  // fluid drag on the core, evaluated in body axes (see qd_model.h on the principal frame);
  // COM at (0,0,c0z): v_com = vb + w x c0
  V3<T> fD0, tD0;
  fluid(M.klin0, M.kang0, M.qlx0, M.qly0, M.qlz0, M.qax0, M.qay0, M.qaz0, w,
        mk<T>(vb.x + w.y * M.c0z, vb.y - w.x * M.c0z, vb.z), &fD0, &tD0);
  // u = w x (w x zhat): velocity-product acceleration per unit height on the body z axis
  const V3<T> u = mk<T>(w.x * w.z, w.y * w.z, -(w.x * w.x + w.y * w.y));
  const T sz = T(Const::sense_z);

  if (!LOAD) {
    // single rigid body: rotate about the COM, then recover the origin acceleration
    const V3<T> ac0 = gt + M.c0z * u;
    const V3<T> fl = fT + fD0 - M.m0 * ac0;
    const V3<T> N0 = mk<T>(w.y * w.z * (M.I0z - M.I0y), w.z * w.x * (M.I0x - M.I0z), w.x * w.y * (M.I0y - M.I0x));
    const V3<T> to = tT + tD0 - N0;  // c0 x thrust = 0 (both along z)
    const V3<T> al = mk<T>(to.x * frcp(M.I0x), to.y * frcp(M.I0y), to.z * frcp(M.I0z));
    const V3<T> a0 = frcp(M.m0) * fl - mk<T>(al.y * M.c0z, -al.x * M.c0z, T(0));
    ex->lin = mul(R, a0); ex->ang = al; ex->thdd1 = ex->thdd2 = T(0);
    *im = *ex;
    *acc = a0 + gt + mk<T>(al.y * sz, -al.x * sz, T(0)) + sz * u;
    return;
  }

  // ---- tether geometry (float32 trigonometry) -------------------------------------------
  T s1, c1, s2, c2;
  qsincos(s.th1, &s1, &c1);
  qsincos(s.th2, &s2, &c2);
  const T az = T(Const::anchor_z);
  const V3<T> d = mk<T>(-s2, s1 * c2, -c1 * c2);   // unit vector anchor -> load (F0 axes)
  const V3<T> y2 = mk<T>(T(0), c1, s1);            // hinge-2 axis (F0 axes); hinge-1 axis is x
  const V3<T> e_x = mk<T>(c2, s1 * s2, -c1 * s2);  // x axis of the tether frame F2 = Rx Ry (its z axis is -d)

  // ---- fluid drag on link (frame F1 = Rx(th1)) and tether (frame F2), float32 ------------
  V3<T> fD1, tD1, fD2, tD2;
  {
    const V3<T> w1 = mk<T>(w.x + s.thd1, w.y, w.z);
    const V3<T> w2 = w1 + s.thd2 * y2;
    const V3<T> va = mk<T>(vb.x + w.y * az, vb.y - w.x * az, vb.z);  // anchor velocity
    const V3<T> vc2 = va + M.lc * cross(w2, d);                      // tether COM velocity
    const T k1 = T(LinkFluid::klin), k2 = T(LinkFluid::kang), k3 = T(LinkFluid::ql), k4 = T(LinkFluid::qa);
    // F0 -> F1 components: Rx^T v = (x, c1 y + s1 z, -s1 y + c1 z)
    V3<T> fl, tl;
    fluid(k1, k2, k3, k3, k3, k4, k4, k4, mk<T>(w1.x, c1 * w1.y + s1 * w1.z, -s1 * w1.y + c1 * w1.z),
          mk<T>(va.x, c1 * va.y + s1 * va.z, -s1 * va.y + c1 * va.z), &fl, &tl);
    fD1 = mk<T>(fl.x, c1 * fl.y - s1 * fl.z, s1 * fl.y + c1 * fl.z);
    tD1 = mk<T>(tl.x, c1 * tl.y - s1 * tl.z, s1 * tl.y + c1 * tl.z);
    fluid(M.klin2, M.kang2, M.qlt2, M.qlt2, M.qla2, M.qat2, M.qat2, M.qaa2,
          mk<T>(dot(e_x, w2), dot(y2, w2), -dot(d, w2)), mk<T>(dot(e_x, vc2), dot(y2, vc2), -dot(d, vc2)), &fl, &tl);
    fD2 = fl.x * e_x + fl.y * y2 - fl.z * d;
    tD2 = tl.x * e_x + tl.y * y2 - tl.z * d;
  }

  // ---- velocity-product terms and generalised forces, HP ---------------------------------
  const HP m0 = M.m0, m1 = Const::m1, m2 = M.m2, i1 = Const::I1, It = M.I2t, lc = M.lc;
  const HP c0z = M.c0z, azh = Const::anchor_z;
  const HP dI = HP(M.I2a) - It;
  const V3<HP> wh = cvt<HP>(w), gth = cvt<HP>(gt), dh = cvt<HP>(d), y2h = cvt<HP>(y2);
  const V3<HP> uh = mk<HP>(wh.x * wh.z, wh.y * wh.z, -(wh.x * wh.x + wh.y * wh.y));
  const HP thd1 = s.thd1, thd2 = s.thd2;
  const V3<HP> rho = lc * dh;                                         // anchor -> tether COM
  const V3<HP> w1 = mk<HP>(wh.x + thd1, wh.y, wh.z);
  const V3<HP> w2 = w1 + thd2 * y2h;
  // accelerations with all generalised accelerations zero and origin acceleration g~
  const V3<HP> aa = gth + azh * uh;                                   // anchor
  const V3<HP> al1 = mk<HP>(HP(0), thd1 * wh.z, -thd1 * wh.y);        // thd1 * (w x xhat)
  const V3<HP> al2 = al1 + thd2 * cross(w1, y2h);
  const V3<HP> ac2 = aa + cross(al2, rho) + dot(w2, rho) * w2 - dot(w2, w2) * rho;
  // inertial wrenches
  const V3<HP> F0 = m0 * (gth + c0z * uh);
  const V3<HP> N0 = mk<HP>(wh.y * wh.z * (HP(M.I0z) - HP(M.I0y)), wh.z * wh.x * (HP(M.I0x) - HP(M.I0z)),
                           wh.x * wh.y * (HP(M.I0y) - HP(M.I0x)));
  const V3<HP> F1 = m1 * aa;
  const V3<HP> N1 = i1 * al1;
  const V3<HP> F2 = m2 * ac2;
  const V3<HP> N2 = It * al2 + (dI * dot(dh, al2)) * dh + (dI * dot(dh, w2)) * cross(w2, dh);
  // applied minus inertial
  const V3<HP> G0 = cvt<HP>(fD0) - F0, G1 = cvt<HP>(fD1) - F1, G2 = cvt<HP>(fD2) - F2;
  const V3<HP> fl = cvt<HP>(fT) + G0 + G1 + G2;
  const V3<HP> W2 = cvt<HP>(tD2) - N2 + cross(rho, G2);                // wrench on the tether about the anchor
  const V3<HP> T1 = cvt<HP>(tD1) - N1;
  const V3<HP> G12 = G1 + G2;
  const V3<HP> fw = cvt<HP>(tT) + cvt<HP>(tD0) - N0 + mk<HP>(-c0z * G0.y, c0z * G0.x, HP(0)) + T1 +
                    mk<HP>(-azh * G12.y, azh * G12.x, HP(0)) + W2;
  const HP bd = Const::damping;
  const HP ft1 = T1.x + W2.x - bd * thd1;
  const HP ft2 = dot(y2h, W2) - bd * thd2;

  // ---- mass matrix about the system COM: sums of non-negative terms ------------------------
  const HP mt = m0 + m1 + m2, imt = frcp(mt);
  const V3<HP> r2 = mk<HP>(rho.x, rho.y, rho.z + azh);
  const V3<HP> S = mk<HP>(m2 * r2.x, m2 * r2.y, m0 * c0z + m1 * azh + m2 * r2.z);
  const V3<HP> rc = imt * S;
  const HP c1h = c1, s1h = s1, c2h = c2, s2h = s2;
  const V3<HP> p1 = lc * mk<HP>(HP(0), c1h * c2h, s1h * c2h);          // xhat x rho
  const V3<HP> p2 = lc * mk<HP>(-c2h, -s1h * s2h, c1h * s2h);          // y2 x rho
  const V3<HP> q2 = r2 - rc;
  const HP qx = -rc.x, qy = -rc.y, q0z = c0z - rc.z, q1z = azh - rc.z; // core and link COM offsets from the COM
  const HP m01 = m0 + m1, base = i1 + It;
  const HP zz01 = m0 * q0z * q0z + m1 * q1z * q1z, z01 = m0 * q0z + m1 * q1z;
  const HP Jxx = HP(M.I0x) + base + dI * dh.x * dh.x + m01 * qy * qy + zz01 + m2 * (q2.y * q2.y + q2.z * q2.z);
  const HP Jyy = HP(M.I0y) + base + dI * dh.y * dh.y + m01 * qx * qx + zz01 + m2 * (q2.x * q2.x + q2.z * q2.z);
  const HP Jzz = HP(M.I0z) + base + dI * dh.z * dh.z + m01 * (qx * qx + qy * qy) + m2 * (q2.x * q2.x + q2.y * q2.y);
  const HP Jxy = dI * dh.x * dh.y - m01 * qx * qy - m2 * q2.x * q2.y;
  const HP Jxz = dI * dh.x * dh.z - qx * z01 - m2 * q2.x * q2.z;
  const HP Jyz = dI * dh.y * dh.z - qy * z01 - m2 * q2.y * q2.z;
  const V3<HP> B1 = mk<HP>(base, HP(0), HP(0)) + (dI * dh.x) * dh + m2 * cross(q2, p1);
  const V3<HP> B2 = It * y2h + m2 * cross(q2, p2);
  const HP mu = m2 * (mt - m2) * imt;
  const HP lc2 = lc * lc;
  const HP D1 = base + dI * s2h * s2h + mu * lc2 * c2h * c2h;
  const HP D2 = It + mu * lc2;
  const V3<HP> fwr = fw - cross(rc, fl);
  const HP k = m2 * imt;
  const HP g1 = ft1 - k * dot(p1, fl);
  const HP g2 = ft2 - k * dot(p2, fl);

  // ---- LDL^T of the 3x3 block, three right-hand sides ----------------------------------------
  const HP d0 = frcp(Jxx);
  const HP l10 = Jxy * d0, l20 = Jxz * d0;
  const HP d1 = frcp(Jyy - l10 * Jxy);
  const HP t21 = Jyz - l20 * Jxy;
  const HP l21 = t21 * d1;
  const HP d2 = frcp(Jzz - l20 * Jxz - l21 * t21);
#define QD_SOLVE3(b, o)                                       \
  {                                                           \
    const HP y0 = (b).x, y1 = (b).y - l10 * y0;               \
    const HP y2_ = (b).z - l20 * y0 - l21 * y1;               \
    const HP z2 = y2_ * d2;                                   \
    const HP z1 = y1 * d1 - l21 * z2;                         \
    const HP z0 = y0 * d0 - l10 * z1 - l20 * z2;              \
    (o) = mk<HP>(z0, z1, z2);                                 \
  }
  V3<HP> Xf, X1, X2;
  QD_SOLVE3(fwr, Xf);
  QD_SOLVE3(B1, X1);
  QD_SOLVE3(B2, X2);
#undef QD_SOLVE3
  // 2x2 Schur complement on the hinges
  const HP s11 = D1 - dot(B1, X1), s12 = -dot(B1, X2), s22 = D2 - dot(B2, X2);
  const HP q1 = g1 - dot(B1, Xf), q2s = g2 - dot(B2, Xf);
  const HP hb = HP(h) * bd;
  V3<HP> a0ex;
#define QD_FINISH(S11, S22, out, A0)                                                     \
  {                                                                                      \
    const HP idet = frcp((S11) * (S22) - s12 * s12);                                     \
    const HP t1 = ((S22) * q1 - s12 * q2s) * idet, t2 = ((S11) * q2s - s12 * q1) * idet; \
    const V3<HP> al = Xf - t1 * X1 - t2 * X2;                                            \
    A0 = imt * (fl - cross(al, S) - (m2 * t1) * p1 - (m2 * t2) * p2);                    \
    (out)->ang = cvt<T>(al); (out)->thdd1 = T(t1); (out)->thdd2 = T(t2);                 \
  }
  V3<HP> a0im;
  QD_FINISH(s11, s22, ex, a0ex);
  QD_FINISH(s11 + hb, s22 + hb, im, a0im);
#undef QD_FINISH
  const V3<T> a0e = cvt<T>(a0ex);
  *acc = a0e + gt + mk<T>(ex->ang.y * sz, -ex->ang.x * sz, T(0)) + sz * u;
  ex->lin = mul(R, a0e);
  im->lin = mul(R, cvt<T>(a0im));
}

// Euler advance of one substep with the accelerations `im`: activations, velocities, then positions with the NEW velocities.
// In two parts because the first does not need the accelerations (the cooperative kernel runs it while it waits for them).
template <class T>
QD_HD void integrate_act(const Model<T>& M, State<T>& s, T c0, T c1, T c2, T c3, T h) {
  // activations: explicit Euler on act_dot = (ctrl - act)/tau, computed from the pre-step act
  const T ht = h * M.inv_tau;
  s.a0 += ht * (c0 - s.a0); s.a1 += ht * (c1 - s.a1); s.a2 += ht * (c2 - s.a2); s.a3 += ht * (c3 - s.a3);
}
// (w, x, y, z): the NORMALISED attitude; (wx, wy, wz): the new body rates.  q <- normalise(q (x) exp(h w / 2)) (mju_quatIntegrate).
// Separate from integrate_motion() because the latency kernel normalises the old quaternion a phase ahead of the solve.
template <class T>
QD_HD void quat_advance(T w, T x, T y, T z, T wx, T wy, T wz, T h, T* ow, T* ox, T* oy, T* oz) {
  const T w2 = wx * wx + wy * wy + wz * wz;
  T ax = T(1), ay = T(0), az = T(0), ang = T(0);
  if (w2 >= T(1e-30)) { const T iw = frsq(w2); ax = wx * iw; ay = wy * iw; az = wz * iw; ang = h * (w2 * iw); }
  T sh, ch;
  qsincos(T(0.5) * ang, &sh, &ch);
  const T rx = ax * sh, ry = ay * sh, rz = az * sh;
  const T nw = w * ch - x * rx - y * ry - z * rz;
  const T nx = w * rx + x * ch + y * rz - z * ry;
  const T ny = w * ry - x * rz + y * ch + z * rx;
  const T nz = w * rz + x * ry - y * rx + z * ch;
  const T qn = frsq(nw * nw + nx * nx + ny * ny + nz * nz);
  *ow = nw * qn; *ox = nx * qn; *oy = ny * qn; *oz = nz * qn;
}
template <class T, bool LOAD>
QD_HD void integrate_motion(State<T>& s, const Accel<T>& im, T h) {
  // velocities, then positions with the NEW velocities
  s.vx += h * im.lin.x; s.vy += h * im.lin.y; s.vz += h * im.lin.z;
  s.wx += h * im.ang.x; s.wy += h * im.ang.y; s.wz += h * im.ang.z;
  s.px += h * s.vx; s.py += h * s.vy; s.pz += h * s.vz;
  if (LOAD) {
    s.thd1 += h * im.thdd1; s.thd2 += h * im.thdd2;
    s.th1 += h * s.thd1; s.th2 += h * s.thd2;
  }
  // quaternion exponential map with the body-frame rate, then renormalise
  {
    const T qn = frsq(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);
    quat_advance(s.qw * qn, s.qx * qn, s.qy * qn, s.qz * qn, s.wx, s.wy, s.wz, h, &s.qw, &s.qx, &s.qy, &s.qz);
  }
}
template <class T, bool LOAD>
QD_HD void integrate(const Model<T>& M, State<T>& s, const Accel<T>& im, T c0, T c1, T c2, T c3, T h) {
  integrate_act(M, s, c0, c1, c2, c3, h);
  integrate_motion<T, LOAD>(s, im, h);
}

// one physics substep (mj_step with nstep = 1): forward, then Euler advance.
// ctrl must already be clamped to [0,1].  Returns the accelerometer reading.
template <class T, bool LOAD>
QD_HD V3<T> substep(const Model<T>& M, State<T>& s, T c0, T c1, T c2, T c3, T h) {
  Accel<T> ex, im;
  V3<T> acc;
  forward<T, LOAD>(M, s, h, &ex, &im, &acc);
  integrate<T, LOAD>(M, s, im, c0, c1, c2, c3, h);
  return acc;
}

}  // namespace qd
