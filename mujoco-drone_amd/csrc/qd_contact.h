// qd_contact.h -- contact of the single-body drone with the floor plane z = 0 (env_gen.py:97: every drone geom
// has contype 1 / conaffinity 0, the floor 1 / 1, so the only pairs are drone geom vs floor).
//
// In the manner of MuJoCo's soft-constraint contact model, PARITY UNPINNED: the contact generation rules
// (plane-box: the corners below the plane that are not above the box centre, at most four; plane-cylinder: the deepest point
// of the near cap's rim, the matching point of the far cap, two more rim points at +-120 degrees) and the constraint constants
// (solref 0.02 / 1, solimp 0.9 / 0.95 / 0.001 / 0.5 / 2, pyramidal friction cone with mu = 1, regulariser of a pyramid edge
// R = 2 mu^2 (1 - d)/d (1 + mu^2) / m) restate MuJoCo's documentation and published source from memory; nothing under
// /root/reference pins them.  What IS checked: the float64 checker of the test suite solves the same convex problem by a different route (projected
// Gauss-Seidel on the dual, general Jacobians, full mass matrix) and the two agree; resting force = weight, critically damped
// settling with the 0.02 s time constant, Coulomb sliding at mu g.
//
// Route here: a Newton method on the primal.  Unknown x = (change of the COM acceleration in world axes, change of the angular
// acceleration in body axes) -- in these coordinates the mass matrix of the single rigid body is diag(m, m, m, Ix, Iy, Iz) --
//   minimise  1/2 x^T M x + sum over pyramid edges e of 1/2 D_e min(0, c_e + J_e x)^2,   D_e = 1 / R_e,
// where c_e = J_e qacc_unconstrained - aref_e and aref_e = -b (J_e qvel) - k d(r) r (MuJoCo's reference acceleration).
// Everything in float64: this path runs only while a geom is below the floor.
#pragma once
#include "qd_dynamics.h"

namespace qd {

constexpr int CONTACT_MAX = 40;  // 14 geoms: at most 6 boxes x 4 + 8 cylinders x 4 = 56; more than 40 at once needs the drone half buried

struct ContactSet {
  int n;
  double x[CONTACT_MAX], y[CONTACT_MAX], z[CONTACT_MAX], r[CONTACT_MAX];  // world position, signed distance (< 0)
  QD_HD void push(double px, double py, double pz, double dist) {
    if (n < CONTACT_MAX) { x[n] = px; y[n] = py; z[n] = pz; r[n] = dist; n++; }
  }
};

// centre c, orientation columns of Rg (row-major 3 x 3), half sizes
QD_HD void contact_box(ContactSet& cs, const double c[3], const double Rg[9], double sx, double sy, double sz) {
  int n = 0;
  for (int i = 0; i < 8 && n < 4; i++) {
    const double vx = (i & 1) ? sx : -sx, vy = (i & 2) ? sy : -sy, vz = (i & 4) ? sz : -sz;
    const double ox = Rg[0] * vx + Rg[1] * vy + Rg[2] * vz, oy = Rg[3] * vx + Rg[4] * vy + Rg[5] * vz, oz = Rg[6] * vx + Rg[7] * vy + Rg[8] * vz;
    if (c[2] + oz > 0.0 || oz > 0.0) continue;
    const double dist = c[2] + oz;
    cs.push(c[0] + ox, c[1] + oy, c[2] + oz - 0.5 * dist, dist);
    n++;
  }
}

QD_HD void contact_cylinder(ContactSet& cs, const double c[3], const double Rg[9], double radius, double hh) {
  double ax = Rg[2], ay = Rg[5], az = Rg[8];
  double prjaxis = az;
  if (prjaxis > 0.0) { ax = -ax; ay = -ay; az = -az; prjaxis = -prjaxis; }
  double vx = ax * prjaxis, vy = ay * prjaxis, vz = az * prjaxis - 1.0;  // -normal projected on the plane of the disk
  const double len = sqrt(vx * vx + vy * vy + vz * vz);
  if (len < 1e-12) { vx = Rg[0] * radius; vy = Rg[3] * radius; vz = Rg[6] * radius; }
  else { const double k = radius / len; vx *= k; vy *= k; vz *= k; }
  const double prjvec = vz;
  ax *= hh; ay *= hh; az *= hh; prjaxis *= hh;
  const double d0 = c[2];
  if (d0 + prjaxis + prjvec > 0.0) return;
  {
    const double d = d0 + prjaxis + prjvec;
    cs.push(c[0] + vx + ax, c[1] + vy + ay, c[2] + vz + az - 0.5 * d, d);
  }
  if (d0 - prjaxis + prjvec <= 0.0) {
    const double d = d0 - prjaxis + prjvec;
    cs.push(c[0] + vx - ax, c[1] + vy - ay, c[2] + vz - az - 0.5 * d, d);
  }
  const double prjvec1 = -0.5 * prjvec;
  if (d0 + prjaxis + prjvec1 <= 0.0) {
    double wx = vy * az - vz * ay, wy = vz * ax - vx * az, wz = vx * ay - vy * ax;  // vec x axis
    const double l1 = sqrt(wx * wx + wy * wy + wz * wz);
    if (l1 > 1e-12) { const double k = radius * 0.86602540378443864676 / l1; wx *= k; wy *= k; wz *= k; }
    const double d = d0 + prjaxis + prjvec1;
    cs.push(c[0] + wx + ax - 0.5 * vx, c[1] + wy + ay - 0.5 * vy, c[2] + wz + az - 0.5 * vz - 0.5 * d, d);
    cs.push(c[0] - wx + ax - 0.5 * vx, c[1] - wy + ay - 0.5 * vy, c[2] - wz + az - 0.5 * vz - 0.5 * d, d);
  }
}

// the 14 geoms of make_drone (env_gen.py:41-61), numbers as they reach MuJoCo (%.5g); p = body origin, R = body -> world
QD_HD void contact_generate(ContactSet& cs, double arm_len, const double p[3], const double R[9]) {
  cs.n = 0;
  const double hb = 0.05, sq2 = 1.4142135623730951, cs45 = 0.70710678118654752440;
  const double pa = round5((sq2 * hb + 0.5 * arm_len) * cs45), pm = round5((sq2 * hb + arm_len) * cs45);
  const double Id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  auto place = [&](double lx, double ly, double lz, const double Rl[9], double c[3], double Rg[9]) {
    c[0] = p[0] + R[0] * lx + R[1] * ly + R[2] * lz;
    c[1] = p[1] + R[3] * lx + R[4] * ly + R[5] * lz;
    c[2] = p[2] + R[6] * lx + R[7] * ly + R[8] * lz;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Rg[3 * i + j] = R[3 * i] * Rl[j] + R[3 * i + 1] * Rl[3 + j] + R[3 * i + 2] * Rl[6 + j];
  };
  double c[3], Rg[9];
  place(0, 0, 0, Id, c, Rg);
  contact_box(cs, c, Rg, round5(hb), round5(hb), round5(hb / 3));
  place(round5(hb + hb / 3), 0, 0, Id, c, Rg);
  contact_box(cs, c, Rg, round5(hb / 3), round5(0.15 * hb), round5(0.15 * hb));
  const double sgx[4] = {1, 1, -1, -1}, sgy[4] = {-1, 1, 1, -1};                       // theta = i pi/2 - pi/4
  const double th[4] = {-round5(0.78539816339744830962), round5(0.78539816339744830962), round5(2.35619449019234492885),
                        round5(3.92699081698724154808)};
  for (int i = 0; i < 4; i++) {
    const double ct = cos(th[i]), st = sin(th[i]);
    const double Rz[9] = {ct, -st, 0, st, ct, 0, 0, 0, 1};
    place(sgx[i] * pa, sgy[i] * pa, 0, Rz, c, Rg);
    contact_box(cs, c, Rg, round5(arm_len / 2), round5(arm_len / 20), round5(arm_len / 20));
    place(sgx[i] * pm, sgy[i] * pm, round5(0.015), Id, c, Rg);
    contact_cylinder(cs, c, Rg, round5(0.01), round5(0.01));
    place(sgx[i] * pm, sgy[i] * pm, round5(0.025), Id, c, Rg);
    contact_cylinder(cs, c, Rg, round5(arm_len / 1.5), round5(0.0025));
  }
}

QD_HD double contact_impedance(double r) {  // solimp (0.9, 0.95, 0.001, 0.5, 2)
  double x = fabs(r) * 1000.0;
  if (x > 1.0) x = 1.0;
  const double y = x <= 0.5 ? 2.0 * x * x : 1.0 - 2.0 * (1.0 - x) * (1.0 - x);
  return 0.9 + 0.05 * y;
}

// 6 x 6 SPD solve (Cholesky), H row-major, overwritten
QD_HD void contact_solve6(double H[36], const double g[6], double dx[6]) {
  for (int i = 0; i < 6; i++)
    for (int j = 0; j <= i; j++) {
      double s = H[6 * i + j];
      for (int k = 0; k < j; k++) s -= H[6 * i + k] * H[6 * j + k];
      H[6 * i + j] = (i == j) ? sqrt(s) : s / H[6 * j + j];
    }
  double y[6];
  for (int i = 0; i < 6; i++) {
    double s = g[i];
    for (int k = 0; k < i; k++) s -= H[6 * i + k] * y[k];
    y[i] = s / H[6 * i + i];
  }
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < 6; k++) s -= H[6 * k + i] * dx[k];
    dx[i] = s / H[6 * i + i];
  }
}

// Adds the floor's reaction to the unconstrained accelerations of the single-body model.
//   lin: world-frame acceleration of the body origin (qacc[0:3]), ang: body-frame angular acceleration (qacc[3:6])
// Returns the number of contacts; *force_z the total normal force.
template <class T>
QD_HD int floor_contact(const Model<T>& M, const State<T>& s, double arm_len, double h, V3<T>& lin, V3<T>& ang, double* force_z) {
  if (force_z) *force_z = 0.0;
  // cheap exit: nothing of the drone reaches further than arm + propeller radius from the origin
  const double reach = 1.4142135623730951 * 0.05 + arm_len * (1.0 + 1.0 / 1.5) + 0.03;
  if ((double)s.pz > reach) return 0;
  const double qn = 1.0 / sqrt((double)s.qw * s.qw + (double)s.qx * s.qx + (double)s.qy * s.qy + (double)s.qz * s.qz);
  const double w = s.qw * qn, x = s.qx * qn, y = s.qy * qn, z = s.qz * qn;
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                       2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                       2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
  const double p[3] = {(double)s.px, (double)s.py, (double)s.pz};
  ContactSet cs;
  contact_generate(cs, arm_len, p, R);
  int nact = 0;
  for (int c = 0; c < cs.n; c++) nact += cs.r[c] < 0.0;
  if (nact == 0) return cs.n;

  const double m = (double)M.m0, I[3] = {(double)M.I0x, (double)M.I0y, (double)M.I0z}, cz = (double)M.c0z;
  const double com[3] = {p[0] + R[2] * cz, p[1] + R[5] * cz, p[2] + R[8] * cz};
  const double v[3] = {(double)s.vx, (double)s.vy, (double)s.vz}, wb[3] = {(double)s.wx, (double)s.wy, (double)s.wz};
  const double ww[3] = {R[0] * wb[0] + R[1] * wb[1] + R[2] * wb[2], R[3] * wb[0] + R[4] * wb[1] + R[5] * wb[2], R[6] * wb[0] + R[7] * wb[1] + R[8] * wb[2]};
  const double a0[3] = {(double)lin.x, (double)lin.y, (double)lin.z}, al0[3] = {(double)ang.x, (double)ang.y, (double)ang.z};
  const double aw[3] = {R[0] * al0[0] + R[1] * al0[1] + R[2] * al0[2], R[3] * al0[0] + R[4] * al0[1] + R[5] * al0[2], R[6] * al0[0] + R[7] * al0[1] + R[8] * al0[2]};
  const double mu = 1.0, tc = h * 2.0 > 0.02 ? h * 2.0 : 0.02, dmax = 0.95;
  const double kb = 2.0 / (dmax * tc), kk = 1.0 / (dmax * dmax * tc * tc);
  const double dir[4][3] = {{0, mu, 1}, {0, -mu, 1}, {-mu, 0, 1}, {mu, 0, 1}};  // contact frame: normal z, tangents y and -x

  // row e of contact c: J = [u ; R^T ((x - com) x u)], residual at x = 0, weight D
  auto row = [&](int c, int e, double J[6], double* res, double* D) {
    const double* u = dir[e];
    const double rc[3] = {cs.x[c] - com[0], cs.y[c] - com[1], cs.z[c] - com[2]};
    const double tw[3] = {rc[1] * u[2] - rc[2] * u[1], rc[2] * u[0] - rc[0] * u[2], rc[0] * u[1] - rc[1] * u[0]};
    J[0] = u[0]; J[1] = u[1]; J[2] = u[2];
    J[3] = R[0] * tw[0] + R[3] * tw[1] + R[6] * tw[2];
    J[4] = R[1] * tw[0] + R[4] * tw[1] + R[7] * tw[2];
    J[5] = R[2] * tw[0] + R[5] * tw[1] + R[8] * tw[2];
    const double ro[3] = {cs.x[c] - p[0], cs.y[c] - p[1], cs.z[c] - p[2]};  // from the body origin: MuJoCo's J acts on qacc / qvel of the origin
    const double vp[3] = {v[0] + ww[1] * ro[2] - ww[2] * ro[1], v[1] + ww[2] * ro[0] - ww[0] * ro[2], v[2] + ww[0] * ro[1] - ww[1] * ro[0]};
    const double ap[3] = {a0[0] + aw[1] * ro[2] - aw[2] * ro[1], a0[1] + aw[2] * ro[0] - aw[0] * ro[2], a0[2] + aw[0] * ro[1] - aw[1] * ro[0]};
    const double imp = contact_impedance(cs.r[c]);
    const double aref = -kb * (u[0] * vp[0] + u[1] * vp[1] + u[2] * vp[2]) - kk * imp * cs.r[c];
    *res = (u[0] * ap[0] + u[1] * ap[1] + u[2] * ap[2]) - aref;
    double Rr = 2.0 * mu * mu * (1.0 - imp) / imp * (1.0 + mu * mu) / m;
    if (Rr < 1e-15) Rr = 1e-15;
    *D = 1.0 / Rr;
  };
  const double Md[6] = {m, m, m, I[0], I[1], I[2]};
  auto cost = [&](const double xx[6]) {
    double cst = 0.0;
    for (int k = 0; k < 6; k++) cst += 0.5 * Md[k] * xx[k] * xx[k];
    for (int c = 0; c < cs.n; c++) {
      if (!(cs.r[c] < 0.0)) continue;
      for (int e = 0; e < 4; e++) {
        double J[6], res, D;
        row(c, e, J, &res, &D);
        double val = res;
        for (int k = 0; k < 6; k++) val += J[k] * xx[k];
        if (val < 0.0) cst += 0.5 * D * val * val;
      }
    }
    return cst;
  };
  double xk[6] = {0, 0, 0, 0, 0, 0};
  double ck = cost(xk);
  for (int it = 0; it < 60; it++) {
    double g[6], H[36];
    for (int k = 0; k < 6; k++) g[k] = Md[k] * xk[k];
    for (int k = 0; k < 36; k++) H[k] = 0.0;
    for (int k = 0; k < 6; k++) H[7 * k] = Md[k];
    for (int c = 0; c < cs.n; c++) {
      if (!(cs.r[c] < 0.0)) continue;
      for (int e = 0; e < 4; e++) {
        double J[6], res, D;
        row(c, e, J, &res, &D);
        double val = res;
        for (int k = 0; k < 6; k++) val += J[k] * xk[k];
        if (val < 0.0) {
          for (int i = 0; i < 6; i++) {
            g[i] += D * val * J[i];
            for (int j = 0; j <= i; j++) H[6 * i + j] += D * J[i] * J[j];
          }
        }
      }
    }
    double gn = 0.0;
    for (int k = 0; k < 6; k++) gn += g[k] * g[k] / Md[k];  // squared gradient in the M^-1 norm: an acceleration squared times mass
    if (gn < 1e-22 * (1.0 + ck)) break;
    double dx[6], ng[6];
    for (int k = 0; k < 6; k++) ng[k] = -g[k];
    contact_solve6(H, ng, dx);
    double slope = 0.0;
    for (int k = 0; k < 6; k++) slope += g[k] * dx[k];
    double t = 1.0, cn = ck;
    double xn[6];
    for (int ls = 0; ls < 30; ls++) {  // backtracking; the full step is exact when the active set does not change
      for (int k = 0; k < 6; k++) xn[k] = xk[k] + t * dx[k];
      cn = cost(xn);
      if (cn <= ck + 1e-4 * t * slope) break;
      t *= 0.5;
    }
    for (int k = 0; k < 6; k++) xk[k] = xn[k];
    if (ck - cn < 1e-16 * (1.0 + fabs(ck))) { ck = cn; break; }
    ck = cn;
  }
  // back to the origin's acceleration: a_origin = a_com - (R alpha) x (R c)
  const double da[3] = {R[0] * xk[3] + R[1] * xk[4] + R[2] * xk[5], R[3] * xk[3] + R[4] * xk[4] + R[5] * xk[5], R[6] * xk[3] + R[7] * xk[4] + R[8] * xk[5]};
  const double rc[3] = {R[2] * cz, R[5] * cz, R[8] * cz};
  lin.x = T(a0[0] + xk[0] - (da[1] * rc[2] - da[2] * rc[1]));
  lin.y = T(a0[1] + xk[1] - (da[2] * rc[0] - da[0] * rc[2]));
  lin.z = T(a0[2] + xk[2] - (da[0] * rc[1] - da[1] * rc[0]));
  ang.x = T(al0[0] + xk[3]); ang.y = T(al0[1] + xk[4]); ang.z = T(al0[2] + xk[5]);
  if (force_z) *force_z = m * xk[2];
  return cs.n;
}

// substep of the single-body model with the floor: forward, the floor's reaction, Euler advance; the accelerometer is
// re-evaluated with the constrained accelerations (MuJoCo computes acceleration sensors after the constraint solve)
template <class T>
QD_HD V3<T> substep_floor(const Model<T>& M, State<T>& s, T c0, T c1, T c2, T c3, T h, double arm_len) {
  Accel<T> ex, im;
  V3<T> acc;
  forward<T, false>(M, s, h, &ex, &im, &acc);
  const V3<T> lin0 = ex.lin, ang0 = ex.ang;
  const int n = floor_contact<T>(M, s, arm_len, (double)h, ex.lin, ex.ang, nullptr);
  if (n > 0) {
    // acc = R^T lin + g~ + alpha x r_s + w x (w x r_s) at the site r_s = (0, 0, sense_z): only the first and third term changed
    const T qn = frsq(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);
    const M3<T> R = quat2mat(s.qw * qn, s.qx * qn, s.qy * qn, s.qz * qn);
    const V3<T> dl = mulT(R, mk<T>(ex.lin.x - lin0.x, ex.lin.y - lin0.y, ex.lin.z - lin0.z));
    const T sz = T(Const::sense_z), dax = ex.ang.x - ang0.x, day = ex.ang.y - ang0.y;
    acc = acc + dl + mk<T>(day * sz, -dax * sz, T(0));
  }
  integrate<T, false>(M, s, ex, c0, c1, c2, c3, h);
  return acc;
}

}  // namespace qd
