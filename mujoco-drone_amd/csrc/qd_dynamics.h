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

// inertia-box fluid wrench for a body with box dims (bx,by,bz), local angular
// velocity w and local COM velocity v (both in the body's inertial-frame axes)
template <class T>
QD_HD void fluid(T bx, T by, T bz, V3<T> w, V3<T> v, V3<T>* f, V3<T>* tq) {
  const T rho = T(Const::density), mu = T(Const::viscosity), pi = T(3.14159265358979323846);
  const T d = (bx + by + bz) * T(1.0 / 3.0);
  const T kang = pi * d * d * d * mu, klin = T(3) * pi * d * mu;
  const T bx2 = bx * bx, by2 = by * by, bz2 = bz * bz;
  const T bx4 = bx2 * bx2, by4 = by2 * by2, bz4 = bz2 * bz2;
  f->x = -(klin + T(0.5) * rho * by * bz * qabs(v.x)) * v.x;
  f->y = -(klin + T(0.5) * rho * bx * bz * qabs(v.y)) * v.y;
  f->z = -(klin + T(0.5) * rho * bx * by * qabs(v.z)) * v.z;
  tq->x = -(kang + rho * bx * (by4 + bz4) * T(1.0 / 64.0) * qabs(w.x)) * w.x;
  tq->y = -(kang + rho * by * (bx4 + bz4) * T(1.0 / 64.0) * qabs(w.y)) * w.y;
  tq->z = -(kang + rho * bz * (bx4 + by4) * T(1.0 / 64.0) * qabs(w.z)) * w.z;
}

template <class T>
QD_HD T boxdim(T Ij, T Ik, T Ii, T mass) {  // sqrt(6 (Ij + Ik - Ii) / mass)
  return qsqrt(qmax(T(1e-15), Ij + Ik - Ii) / mass * T(6));
}

// forward dynamics at the current state.
//   ex  : accelerations with damping explicit (what MuJoCo stores in qacc; feeds the sensor)
//   im  : accelerations of the damping-implicit Euler update ((M + h D) a = M qacc)
//   acc : accelerometer reading (site frame = body frame)
template <class T, bool LOAD>
QD_HD void forward(const Model<T>& M, const State<T>& s, T h, Accel<T>* ex, Accel<T>* im, V3<T>* acc) {
  // attitude (MuJoCo normalises the stored quaternion before use)
  T qn = T(1) / qsqrt(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);
  const M3<T> R = quat2mat(s.qw * qn, s.qx * qn, s.qy * qn, s.qz * qn);
  const V3<T> w = mk<T>(s.wx, s.wy, s.wz);
  const V3<T> vb = mulT(R, mk<T>(s.vx, s.vy, s.vz));  // origin velocity in body axes
  const T g = T(Const::gravity);
  const V3<T> gt = mk<T>(g * R.m20, g * R.m21, g * R.m22);

  // ---- body 0 -------------------------------------------------------------
  const V3<T> c0 = mk<T>(T(0), T(0), M.c0z);
  const V3<T> wxc0 = cross(w, c0);
  const V3<T> ac0 = gt + cross(w, wxc0);
  const V3<T> F0 = M.m0 * ac0;
  const V3<T> N0 = cross(w, mk<T>(M.I0x * w.x, M.I0y * w.y, M.I0z * w.z));
  // rotors (env_gen.py:53-64): thrust along body z at (+-rot, +-rot, 0), yaw reaction +-gearT
  const T f0 = M.gearF * s.a0, f1 = M.gearF * s.a1, f2 = M.gearF * s.a2, f3 = M.gearF * s.a3;
  const V3<T> fT = mk<T>(T(0), T(0), f0 + f1 + f2 + f3);
  const V3<T> tT = mk<T>(M.rot * (-f0 + f1 + f2 - f3), M.rot * (-f0 - f1 + f2 + f3),
                         M.gearT * (s.a0 - s.a1 + s.a2 - s.a3));
  // fluid drag on the core, evaluated in body axes (see qd_model.h on the principal frame)
  V3<T> fD0, tD0;
  fluid(boxdim(M.I0y, M.I0z, M.I0x, M.m0), boxdim(M.I0x, M.I0z, M.I0y, M.m0), boxdim(M.I0x, M.I0y, M.I0z, M.m0), w,
        vb + wxc0, &fD0, &tD0);
  const V3<T> sns = mk<T>(T(0), T(0), T(Const::sense_z));
  const V3<T> sens_vel = cross(w, cross(w, sns));  // w x (w x s)

  if (!LOAD) {
    // single rigid body: rotate about the COM, then recover the origin acceleration
    const V3<T> fl = fT + fD0 - F0;
    const V3<T> to = tT + tD0 - N0;  // c0 x thrust = 0 (both along z)
    const V3<T> al = mk<T>(to.x / M.I0x, to.y / M.I0y, to.z / M.I0z);
    const V3<T> a0 = (T(1) / M.m0) * fl - cross(al, c0);
    ex->lin = mul(R, a0); ex->ang = al; ex->thdd1 = ex->thdd2 = T(0);
    *im = *ex;
    *acc = a0 + gt + cross(al, sns) + sens_vel;
    return;
  }

  // ---- tether geometry ------------------------------------------------------
  T s1, c1, s2, c2;
  qsincos(s.th1, &s1, &c1);
  qsincos(s.th2, &s2, &c2);
  const T m1 = T(Const::m1), i1 = T(Const::I1);
  const V3<T> a = mk<T>(T(0), T(0), T(Const::anchor_z));
  const V3<T> d = mk<T>(-s2, s1 * c2, -c1 * c2);   // unit vector anchor -> load (F0 axes)
  const V3<T> y2 = mk<T>(T(0), c1, s1);            // hinge-2 axis (F0 axes); hinge-1 axis is x
  const V3<T> rho = M.lc * d;                      // anchor -> tether COM
  const T dI = M.I2a - M.I2t;

  // velocities
  const V3<T> w1 = mk<T>(w.x + s.thd1, w.y, w.z);
  const V3<T> w2 = w1 + s.thd2 * y2;
  // velocity-product accelerations (all generalised accelerations zero, origin accel = g~)
  const V3<T> wxa = cross(w, a);
  const V3<T> aa = gt + cross(w, wxa);
  const V3<T> al1 = mk<T>(T(0), s.thd1 * w.z, -s.thd1 * w.y);   // thd1 * (w x xhat)
  const V3<T> al2 = al1 + s.thd2 * cross(w1, y2);
  const V3<T> w2xr = cross(w2, rho);
  const V3<T> ac2 = aa + cross(al2, rho) + cross(w2, w2xr);
  // inertial wrenches
  const V3<T> F1 = m1 * aa;
  const V3<T> N1 = i1 * al1;
  const V3<T> F2 = M.m2 * ac2;
  const T dw = dot(d, w2), da = dot(d, al2);
  const V3<T> N2 = M.I2t * al2 + (dI * da) * d + (dI * dw) * cross(w2, d);

  // fluid drag on link (frame F1 = Rx(th1)) and tether (frame F2 = Rx(th1) Ry(th2))
  const V3<T> va = vb + wxa;       // anchor velocity
  const V3<T> vc2 = va + w2xr;     // tether COM velocity
  V3<T> fD1, tD1, fD2, tD2;
  {
    const T b1 = boxdim(i1, i1, i1, m1);
    // F0 -> F1 components: Rx^T v = (x, c1 y + s1 z, -s1 y + c1 z)
    V3<T> wl = mk<T>(w1.x, c1 * w1.y + s1 * w1.z, -s1 * w1.y + c1 * w1.z);
    V3<T> vl = mk<T>(va.x, c1 * va.y + s1 * va.z, -s1 * va.y + c1 * va.z);
    V3<T> fl, tl;
    fluid(b1, b1, b1, wl, vl, &fl, &tl);
    fD1 = mk<T>(fl.x, c1 * fl.y - s1 * fl.z, s1 * fl.y + c1 * fl.z);
    tD1 = mk<T>(tl.x, c1 * tl.y - s1 * tl.z, s1 * tl.y + c1 * tl.z);
  }
  {
    // E = Rx Ry: columns ex = (c2, s1 s2, -c1 s2), ey = (0, c1, s1) = y2, ez = (s2, -s1 c2, c1 c2) = -d
    const V3<T> e_x = mk<T>(c2, s1 * s2, -c1 * s2);
    const T bt = boxdim(M.I2t, M.I2a, M.I2t, M.m2);   // x and y dims
    const T ba = boxdim(M.I2t, M.I2t, M.I2a, M.m2);   // along the tether
    V3<T> wl = mk<T>(dot(e_x, w2), dot(y2, w2), -dot(d, w2));
    V3<T> vl = mk<T>(dot(e_x, vc2), dot(y2, vc2), -dot(d, vc2));
    V3<T> fl, tl;
    fluid(bt, bt, ba, wl, vl, &fl, &tl);
    fD2 = fl.x * e_x + fl.y * y2 - fl.z * d;
    tD2 = tl.x * e_x + tl.y * y2 - tl.z * d;
  }

  // ---- generalised forces (applied minus velocity-product inertial) -----------
  const V3<T> r2 = a + rho;
  const V3<T> fl = fT + fD0 + fD1 + fD2 - (F0 + F1 + F2);
  const V3<T> W2 = tD2 - N2 + cross(rho, fD2 - F2);   // wrench on the tether about the anchor
  const V3<T> fw = tT + tD0 - N0 + cross(c0, fD0 - F0) + (tD1 - N1) + cross(a, (fD1 - F1) + (fD2 - F2)) + W2;
  const T bd = T(Const::damping);
  const T ft1 = (tD1.x - N1.x) + W2.x - bd * s.thd1;
  const T ft2 = dot(y2, W2) - bd * s.thd2;

  // ---- mass matrix, reduced about the system COM -------------------------------
  const T mt = M.m0 + m1 + M.m2, imt = T(1) / mt;
  const V3<T> S = M.m0 * c0 + m1 * a + M.m2 * r2;
  const V3<T> rc = imt * S;
  const V3<T> p1 = M.lc * mk<T>(T(0), c1 * c2, s1 * c2);        // xhat x rho
  const V3<T> p2 = M.lc * mk<T>(-c2, -s1 * s2, c1 * s2);        // y2 x rho
  // inertia about the origin, then shifted to the COM (symmetric: xx,yy,zz,xy,xz,yz)
  const T c0n = M.c0z * M.c0z, an = T(Const::anchor_z * Const::anchor_z), r2n = dot(r2, r2), rcn = dot(rc, rc);
  const T diag = M.m0 * c0n + i1 + m1 * an + M.I2t + M.m2 * r2n - mt * rcn;
  T Jxx = M.I0x + diag + dI * d.x * d.x - M.m2 * r2.x * r2.x + mt * rc.x * rc.x;
  T Jyy = M.I0y + diag + dI * d.y * d.y - M.m2 * r2.y * r2.y + mt * rc.y * rc.y;
  T Jzz = M.I0z + diag - M.m0 * c0n - m1 * an + dI * d.z * d.z - M.m2 * r2.z * r2.z + mt * rc.z * rc.z;
  T Jxy = dI * d.x * d.y - M.m2 * r2.x * r2.y + mt * rc.x * rc.y;
  T Jxz = dI * d.x * d.z - M.m2 * r2.x * r2.z + mt * rc.x * rc.z;
  T Jyz = dI * d.y * d.z - M.m2 * r2.y * r2.z + mt * rc.y * rc.z;
  const V3<T> rr = r2 - rc;
  const V3<T> B1 = mk<T>(i1 + M.I2t, T(0), T(0)) + (dI * d.x) * d + M.m2 * cross(rr, p1);
  const V3<T> B2 = M.I2t * y2 + M.m2 * cross(rr, p2);
  const T mu = M.m2 * (mt - M.m2) * imt;
  const T lc2 = M.lc * M.lc;
  const T D1 = i1 + M.I2t + dI * s2 * s2 + mu * lc2 * c2 * c2;
  const T D2 = M.I2t + mu * lc2;
  const V3<T> fwr = fw - cross(rc, fl);
  const T k = M.m2 * imt;
  const T g1 = ft1 - k * dot(p1, fl);
  const T g2 = ft2 - k * dot(p2, fl);

  // ---- LDL^T of the 3x3 block, three right-hand sides ---------------------------
  const T d0 = T(1) / Jxx;
  const T l10 = Jxy * d0, l20 = Jxz * d0;
  const T e1 = Jyy - l10 * Jxy;
  const T d1 = T(1) / e1;
  const T t21 = Jyz - l20 * Jxy;
  const T l21 = t21 * d1;
  const T e2 = Jzz - l20 * Jxz - l21 * t21;
  const T d2 = T(1) / e2;
#define QD_SOLVE3(b, o)                                       \
  {                                                           \
    T y0 = (b).x, y1 = (b).y - l10 * y0;                      \
    T y2_ = (b).z - l20 * y0 - l21 * y1;                      \
    T z2 = y2_ * d2;                                          \
    T z1 = y1 * d1 - l21 * z2;                                \
    T z0 = y0 * d0 - l10 * z1 - l20 * z2;                     \
    (o) = mk<T>(z0, z1, z2);                                  \
  }
  V3<T> Xf, X1, X2;
  QD_SOLVE3(fwr, Xf);
  QD_SOLVE3(B1, X1);
  QD_SOLVE3(B2, X2);
#undef QD_SOLVE3
  // 2x2 Schur complement on the hinges
  const T s11 = D1 - dot(B1, X1), s12 = -dot(B1, X2), s22 = D2 - dot(B2, X2);
  const T q1 = g1 - dot(B1, Xf), q2 = g2 - dot(B2, Xf);
  const T hb = h * bd;
#define QD_FINISH(S11, S22, out)                                                       \
  {                                                                                    \
    const T idet = T(1) / ((S11) * (S22) - s12 * s12);                                 \
    const T t1 = ((S22) * q1 - s12 * q2) * idet, t2 = ((S11) * q2 - s12 * q1) * idet;  \
    const V3<T> al = Xf - t1 * X1 - t2 * X2;                                           \
    const V3<T> a0 = imt * (fl - cross(al, S) - (M.m2 * t1) * p1 - (M.m2 * t2) * p2);  \
    (out)->ang = al; (out)->thdd1 = t1; (out)->thdd2 = t2;                             \
    (out)->lin = a0; /* body axes for now */                                           \
  }
  QD_FINISH(s11, s22, ex);
  QD_FINISH(s11 + hb, s22 + hb, im);
#undef QD_FINISH
  *acc = ex->lin + gt + cross(ex->ang, sns) + sens_vel;
  ex->lin = mul(R, ex->lin);
  im->lin = mul(R, im->lin);
}

// one physics substep (mj_step with nstep = 1): forward, then Euler advance.
// ctrl must already be clamped to [0,1].  Returns the accelerometer reading.
template <class T, bool LOAD>
QD_HD V3<T> substep(const Model<T>& M, State<T>& s, T c0, T c1, T c2, T c3, T h) {
  Accel<T> ex, im;
  V3<T> acc;
  forward<T, LOAD>(M, s, h, &ex, &im, &acc);
  // activations: explicit Euler on act_dot = (ctrl - act)/tau, computed from the pre-step act
  const T ht = h * M.inv_tau;
  s.a0 += ht * (c0 - s.a0); s.a1 += ht * (c1 - s.a1); s.a2 += ht * (c2 - s.a2); s.a3 += ht * (c3 - s.a3);
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
    T qn = T(1) / qsqrt(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);
    const T w = s.qw * qn, x = s.qx * qn, y = s.qy * qn, z = s.qz * qn;
    const T wn = qsqrt(s.wx * s.wx + s.wy * s.wy + s.wz * s.wz);
    T ax = T(1), ay = T(0), az = T(0), ang = T(0);
    if (wn >= T(1e-15)) { const T iw = T(1) / wn; ax = s.wx * iw; ay = s.wy * iw; az = s.wz * iw; ang = h * wn; }
    T sh, ch;
    qsincos(T(0.5) * ang, &sh, &ch);
    const T rx = ax * sh, ry = ay * sh, rz = az * sh;
    T nw = w * ch - x * rx - y * ry - z * rz;
    T nx = w * rx + x * ch + y * rz - z * ry;
    T ny = w * ry - x * rz + y * ch + z * rx;
    T nz = w * rz + x * ry - y * rx + z * ch;
    qn = T(1) / qsqrt(nw * nw + nx * nx + ny * ny + nz * nz);
    s.qw = nw * qn; s.qx = nx * qn; s.qy = ny * qn; s.qz = nz * qn;
  }
  return acc;
}

}  // namespace qd
