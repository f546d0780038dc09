// qd_contact.h -- contact of the drone (airframe; with the load also link, tether rod and load box) with the floor plane z = 0 (env_gen.py:97: every drone geom
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
// Route here: a Newton method on the primal.  Single body (floor_contact): unknown x = (change of the COM acceleration in world axes, change of the angular
// acceleration in body axes) -- in these coordinates the mass matrix of the single rigid body is diag(m, m, m, Ix, Iy, Iz) --
//   minimise  1/2 x^T M x + sum over pyramid edges e of 1/2 D_e min(0, c_e + J_e x)^2,   D_e = 1 / R_e,
// where c_e = J_e qacc_unconstrained - aref_e and aref_e = -b (J_e qvel) - k d(r) r (MuJoCo's reference acceleration).
// Everything in float64: this path runs only while a geom is below the floor.
#pragma once
#include "qd_dynamics.h"

// the contact solve is a real call on the device: its registers and scratch are then allocated apart from the flight path's
#define QD_NOINLINE __attribute__((noinline))

namespace qd {

constexpr int CONTACT_MAX = 40;  // 14 geoms: at most 6 boxes x 4 + 8 cylinders x 4 = 56; more than 40 at once needs the drone half buried

struct ContactSet {
  int n, cur;                                                             // cur: body of the geom being tested (0 core, 1 link, 2 tether + load)
  double x[CONTACT_MAX], y[CONTACT_MAX], z[CONTACT_MAX], r[CONTACT_MAX];  // world position, signed distance (< 0)
  signed char b[CONTACT_MAX];
  QD_HD void push(double px, double py, double pz, double dist) {
    if (n < CONTACT_MAX) { x[n] = px; y[n] = py; z[n] = pz; r[n] = dist; b[n] = (signed char)cur; n++; }
  }
};

// centre c, orientation columns of Rg (row-major 3 x 3), half sizes
// CS: anything with push(x, y, z, dist) -- the per-lane ContactSet here, the counting / LDS sinks of qd_contact_group.h
template <class CS>
QD_HD void contact_box(CS& cs, const double c[3], const double Rg[9], double sx, double sy, double sz) {
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

template <class CS>
QD_HD void contact_cylinder(CS& cs, const double c[3], const double Rg[9], double radius, double hh) {
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
  cs.n = 0; cs.cur = 0;
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
QD_HD QD_NOINLINE int floor_contact(const Model<T>& M, const State<T>& s, double arm_len, double h, V3<T>& lin, V3<T>& ang, double* force_z) {
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


// ================================================================================================================================
// The drone + tether + load tree (nv = 8) in MuJoCo's generalised coordinates: qvel = (world linear velocity of the core origin,
// body-frame angular velocity, hinge-x rate, hinge-y rate).  Same convex problem, dense 8 x 8 mass matrix assembled here from the
// three bodies' COM Jacobians (the specialised dynamics of qd_dynamics.h never forms it).
struct TreePose {
  double p[3], R[9], R1[9], R2[9], xa[3];  // core origin / attitude, link frame R Rx(th1), tether frame R1 Ry(th2), hinge anchor
};
QD_HD void tree_mul(const double A[9], const double B[9], double O[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) O[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
QD_HD void tree_pose(const double p[3], const double q[4], double th1, double th2, TreePose& P) {
  const double n = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] * n, x = q[1] * n, y = q[2] * n, z = q[3] * n;
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                       2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
  const double c1 = cos(th1), s1 = sin(th1), c2 = cos(th2), s2 = sin(th2);
  const double Rx[9] = {1, 0, 0, 0, c1, -s1, 0, s1, c1}, Ry[9] = {c2, 0, s2, 0, 1, 0, -s2, 0, c2};
  for (int k = 0; k < 9; k++) P.R[k] = R[k];
  for (int k = 0; k < 3; k++) P.p[k] = p[k];
  tree_mul(P.R, Rx, P.R1);
  tree_mul(P.R1, Ry, P.R2);
  for (int k = 0; k < 3; k++) P.xa[k] = p[k] + R[3 * k + 2] * Const::anchor_z;
}
// 3 x 8 Jacobian of a world point x attached to body b
QD_HD void tree_point_jac(const TreePose& P, int b, const double x[3], double J[3][8]) {
  for (int k = 0; k < 3; k++)
    for (int j = 0; j < 8; j++) J[k][j] = (j == k) ? 1.0 : 0.0;
  const double r[3] = {x[0] - P.p[0], x[1] - P.p[1], x[2] - P.p[2]};
  for (int j = 0; j < 3; j++) {
    const double a[3] = {P.R[j], P.R[3 + j], P.R[6 + j]};
    J[0][3 + j] = a[1] * r[2] - a[2] * r[1]; J[1][3 + j] = a[2] * r[0] - a[0] * r[2]; J[2][3 + j] = a[0] * r[1] - a[1] * r[0];
  }
  if (b >= 1) {
    const double ra[3] = {x[0] - P.xa[0], x[1] - P.xa[1], x[2] - P.xa[2]};
    const double a1[3] = {P.R[0], P.R[3], P.R[6]};
    J[0][6] = a1[1] * ra[2] - a1[2] * ra[1]; J[1][6] = a1[2] * ra[0] - a1[0] * ra[2]; J[2][6] = a1[0] * ra[1] - a1[1] * ra[0];
    if (b == 2) {
      const double a2[3] = {P.R1[1], P.R1[4], P.R1[7]};
      J[0][7] = a2[1] * ra[2] - a2[2] * ra[1]; J[1][7] = a2[2] * ra[0] - a2[0] * ra[2]; J[2][7] = a2[0] * ra[1] - a2[1] * ra[0];
    }
  }
}
// body COMs of the tree
template <class T>
QD_HD void tree_com(const Model<T>& M, const TreePose& P, int b, double c[3]) {
  if (b == 0) for (int k = 0; k < 3; k++) c[k] = P.p[k] + P.R[3 * k + 2] * (double)M.c0z;
  else if (b == 1) for (int k = 0; k < 3; k++) c[k] = P.xa[k];
  else for (int k = 0; k < 3; k++) c[k] = P.xa[k] - P.R2[3 * k + 2] * (double)M.lc;
}
template <class T>
QD_HD void tree_mass_matrix(const Model<T>& M, const TreePose& P, double Mm[64]) {
  for (int k = 0; k < 64; k++) Mm[k] = 0.0;
  const double mass[3] = {(double)M.m0, Const::m1, (double)M.m2};
  const double Id[3][3] = {{(double)M.I0x, (double)M.I0y, (double)M.I0z}, {Const::I1, Const::I1, Const::I1}, {(double)M.I2t, (double)M.I2t, (double)M.I2a}};
  for (int b = 0; b < 3; b++) {
    double c[3], Jp[3][8];
    tree_com(M, P, b, c);
    tree_point_jac(P, b, c, Jp);
    const double* Rb = b == 0 ? P.R : b == 1 ? P.R1 : P.R2;
    // rotational Jacobian in the body's own axes: Rb^T (world axis of each generalised speed)
    double Jw[3][8];
    for (int k = 0; k < 3; k++)
      for (int j = 0; j < 8; j++) Jw[k][j] = 0.0;
    auto setcol = [&](int j, double ax, double ay, double az) {
      for (int k = 0; k < 3; k++) Jw[k][j] = Rb[k] * ax + Rb[3 + k] * ay + Rb[6 + k] * az;
    };
    for (int j = 0; j < 3; j++) setcol(3 + j, P.R[j], P.R[3 + j], P.R[6 + j]);
    if (b >= 1) setcol(6, P.R[0], P.R[3], P.R[6]);
    if (b == 2) setcol(7, P.R1[1], P.R1[4], P.R1[7]);
    for (int i = 0; i < 8; i++)
      for (int j = 0; j <= i; j++) {
        double v = 0.0;
        for (int k = 0; k < 3; k++) v += mass[b] * Jp[k][i] * Jp[k][j] + Id[b][k] * Jw[k][i] * Jw[k][j];
        Mm[8 * i + j] += v;
      }
  }
  for (int i = 0; i < 8; i++)
    for (int j = i + 1; j < 8; j++) Mm[8 * i + j] = Mm[8 * j + i];
}
// Cholesky factor (lower, in place, stride 8) and solve
QD_HD void tree_chol(double A[64]) {
  for (int i = 0; i < 8; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[8 * i + j];
      for (int k = 0; k < j; k++) s -= A[8 * i + k] * A[8 * j + k];
      A[8 * i + j] = (i == j) ? sqrt(s) : s / A[8 * j + j];
    }
}
QD_HD void tree_chol_solve(const double L[64], const double b[8], double x[8]) {
  double y[8];
  for (int i = 0; i < 8; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[8 * i + k] * y[k];
    y[i] = s / L[8 * i + i];
  }
  for (int i = 7; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < 8; k++) s -= L[8 * k + i] * x[k];
    x[i] = s / L[8 * i + i];
  }
}

// Adds the floor's reaction to the accelerations of the load model: `ex` (explicit, what the accelerometer sees) and `im` (with
// the hinge damping implicit, what the Euler step uses: (M + h D) qimp = M qacc like MuJoCo's Euler integrator).
template <class T>
QD_HD QD_NOINLINE int floor_contact_tree(const Model<T>& M, const State<T>& s, double arm_len, double pend_len, double weight_mass, double h,
                             Accel<T>& ex, Accel<T>& im, double* force_z) {
  if (force_z) *force_z = 0.0;
  const double bs = round5(0.1 * cbrt(weight_mass));
  const double reach = 1.4142135623730951 * 0.05 + arm_len * (1.0 + 1.0 / 1.5) + 0.03 + pend_len + 1.7320508075688772 * bs;
  if ((double)s.pz > reach) return 0;
  TreePose P;
  {
    const double p[3] = {(double)s.px, (double)s.py, (double)s.pz}, q[4] = {(double)s.qw, (double)s.qx, (double)s.qy, (double)s.qz};
    tree_pose(p, q, (double)s.th1, (double)s.th2, P);
  }
  ContactSet cs;
  contact_generate(cs, arm_len, P.p, P.R);
  {
    cs.cur = 1;  // link sphere at the anchor
    const double r1 = round5(Const::r1), dist = P.xa[2] - r1;
    if (dist <= 0.0) cs.push(P.xa[0], P.xa[1], P.xa[2] - r1 - 0.5 * dist, dist);
    cs.cur = 2;  // tether rod (cylinder) and load (box), both on the tether frame's z axis
    double c[3];
    const double rz = round5(-pend_len / 2), wz = round5(-pend_len);
    for (int k = 0; k < 3; k++) c[k] = P.xa[k] + P.R2[3 * k + 2] * rz;
    contact_cylinder(cs, c, P.R2, round5(0.005), round5(pend_len / 2));
    for (int k = 0; k < 3; k++) c[k] = P.xa[k] + P.R2[3 * k + 2] * wz;
    contact_box(cs, c, P.R2, bs, bs, bs);
  }
  int nact = 0;
  for (int c = 0; c < cs.n; c++) nact += cs.r[c] < 0.0;
  if (nact == 0) return cs.n;

  double Mm[64];
  tree_mass_matrix(M, P, Mm);
  // body_invweight0 (translational part): J M^-1 J^T of each body's COM at qpos0
  double tran[3];
  {
    TreePose P0;
    const double z3[3] = {0, 0, 0}, q0[4] = {1, 0, 0, 0};
    tree_pose(z3, q0, 0.0, 0.0, P0);
    double L0[64];
    tree_mass_matrix(M, P0, L0);
    tree_chol(L0);
    for (int b = 0; b < 3; b++) {
      double c[3], Jp[3][8], t = 0.0;
      tree_com(M, P0, b, c);
      tree_point_jac(P0, b, c, Jp);
      for (int k = 0; k < 3; k++) {
        double xs[8];
        tree_chol_solve(L0, Jp[k], xs);
        for (int i = 0; i < 8; i++) t += Jp[k][i] * xs[i];
      }
      tran[b] = t / 3.0;
    }
  }
  const double qv[8] = {(double)s.vx, (double)s.vy, (double)s.vz, (double)s.wx, (double)s.wy, (double)s.wz, (double)s.thd1, (double)s.thd2};
  const double a0[8] = {(double)ex.lin.x, (double)ex.lin.y, (double)ex.lin.z, (double)ex.ang.x, (double)ex.ang.y, (double)ex.ang.z,
                        (double)ex.thdd1, (double)ex.thdd2};
  const double mu = 1.0, tc = h * 2.0 > 0.02 ? h * 2.0 : 0.02, dmax = 0.95;
  const double kb = 2.0 / (dmax * tc), kk = 1.0 / (dmax * dmax * tc * tc);
  const double dir[4][3] = {{0, mu, 1}, {0, -mu, 1}, {-mu, 0, 1}, {mu, 0, 1}};
  auto row = [&](int c, int e, double J[8], double* res, double* D) {
    const double xp[3] = {cs.x[c], cs.y[c], cs.z[c]};
    double Jp[3][8];
    tree_point_jac(P, cs.b[c], xp, Jp);
    double vel = 0.0, acc0 = 0.0;
    for (int i = 0; i < 8; i++) {
      J[i] = dir[e][0] * Jp[0][i] + dir[e][1] * Jp[1][i] + dir[e][2] * Jp[2][i];
      vel += J[i] * qv[i];
      acc0 += J[i] * a0[i];
    }
    const double imp = contact_impedance(cs.r[c]);
    *res = acc0 - (-kb * vel - kk * imp * cs.r[c]);
    double Rr = 2.0 * mu * mu * (1.0 - imp) / imp * (1.0 + mu * mu) * tran[cs.b[c]];
    if (Rr < 1e-15) Rr = 1e-15;
    *D = 1.0 / Rr;
  };
  auto cost = [&](const double xx[8]) {
    double cst = 0.0;
    for (int i = 0; i < 8; i++) {
      double mx = 0.0;
      for (int j = 0; j < 8; j++) mx += Mm[8 * i + j] * xx[j];
      cst += 0.5 * xx[i] * mx;
    }
    for (int c = 0; c < cs.n; c++) {
      if (!(cs.r[c] < 0.0)) continue;
      for (int e = 0; e < 4; e++) {
        double J[8], res, D;
        row(c, e, J, &res, &D);
        double val = res;
        for (int k = 0; k < 8; k++) val += J[k] * xx[k];
        if (val < 0.0) cst += 0.5 * D * val * val;
      }
    }
    return cst;
  };
  double xk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double ck = cost(xk), fz = 0.0;
  for (int it = 0; it < 80; it++) {
    double g[8], H[64];
    for (int i = 0; i < 8; i++) {
      g[i] = 0.0;
      for (int j = 0; j < 8; j++) { g[i] += Mm[8 * i + j] * xk[j]; H[8 * i + j] = Mm[8 * i + j]; }
    }
    fz = 0.0;
    for (int c = 0; c < cs.n; c++) {
      if (!(cs.r[c] < 0.0)) continue;
      for (int e = 0; e < 4; e++) {
        double J[8], res, D;
        row(c, e, J, &res, &D);
        double val = res;
        for (int k = 0; k < 8; k++) val += J[k] * xk[k];
        if (val < 0.0) {
          fz -= D * val;  // the edge's force; every edge has a unit normal component
          for (int i = 0; i < 8; i++) {
            g[i] += D * val * J[i];
            for (int j = 0; j <= i; j++) H[8 * i + j] += D * J[i] * J[j];
          }
        }
      }
    }
    double gn = 0.0;
    for (int k = 0; k < 8; k++) gn += g[k] * g[k] / Mm[9 * k];
    if (gn < 1e-22 * (1.0 + ck)) break;
    tree_chol(H);
    double dx[8], ng[8];
    for (int k = 0; k < 8; k++) ng[k] = -g[k];
    tree_chol_solve(H, ng, dx);
    double slope = 0.0;
    for (int k = 0; k < 8; k++) slope += g[k] * dx[k];
    double t = 1.0, cn = ck, xn[8];
    for (int ls = 0; ls < 30; ls++) {
      for (int k = 0; k < 8; k++) xn[k] = xk[k] + t * dx[k];
      cn = cost(xn);
      if (cn <= ck + 1e-4 * t * slope) break;
      t *= 0.5;
    }
    for (int k = 0; k < 8; k++) xk[k] = xn[k];
    if (ck - cn < 1e-16 * (1.0 + fabs(ck))) { ck = cn; break; }
    ck = cn;
  }
  if (force_z) *force_z = fz;
  if (!(fz > 0.0)) return cs.n;  // touching geometry but no force (separating): leave the accelerations exactly as they were
  double qa[8];
  for (int k = 0; k < 8; k++) qa[k] = a0[k] + xk[k];
  ex.lin = mk<T>(T(qa[0]), T(qa[1]), T(qa[2])); ex.ang = mk<T>(T(qa[3]), T(qa[4]), T(qa[5])); ex.thdd1 = T(qa[6]); ex.thdd2 = T(qa[7]);
  // Euler with the hinge damping implicit: (M + h D) qimp = M qacc
  double rt[8], Mh[64], qi[8];
  for (int i = 0; i < 8; i++) {
    rt[i] = 0.0;
    for (int j = 0; j < 8; j++) { rt[i] += Mm[8 * i + j] * qa[j]; Mh[8 * i + j] = Mm[8 * i + j]; }
  }
  Mh[8 * 6 + 6] += h * Const::damping;
  Mh[8 * 7 + 7] += h * Const::damping;
  tree_chol(Mh);
  tree_chol_solve(Mh, rt, qi);
  im.lin = mk<T>(T(qi[0]), T(qi[1]), T(qi[2])); im.ang = mk<T>(T(qi[3]), T(qi[4]), T(qi[5])); im.thdd1 = T(qi[6]); im.thdd2 = T(qi[7]);
  return cs.n;
}

// substep with the floor: forward, the floor's reaction, Euler advance; the accelerometer is re-evaluated with the constrained
// accelerations (MuJoCo computes acceleration sensors after the constraint solve).  raw = the env's six float64 parameters.
template <class T, bool LOAD>
QD_HD V3<T> substep_floor(const Model<T>& M, State<T>& s, T c0, T c1, T c2, T c3, T h, double arm_len, double pend_len,
                          double weight_mass) {
  Accel<T> ex, im;
  V3<T> acc;
  forward<T, LOAD>(M, s, h, &ex, &im, &acc);
  const V3<T> lin0 = ex.lin, ang0 = ex.ang;
  double fz = 0.0;
  if (LOAD) floor_contact_tree<T>(M, s, arm_len, pend_len, weight_mass, (double)h, ex, im, &fz);
  else { floor_contact<T>(M, s, arm_len, (double)h, ex.lin, ex.ang, &fz); im = ex; }
  if (fz > 0.0) {
    // acc = R^T lin + g~ + alpha x r_s + w x (w x r_s) at the site r_s = (0, 0, sense_z): only the first and third term changed
    const T qn = frsq(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);
    const M3<T> R = quat2mat(s.qw * qn, s.qx * qn, s.qy * qn, s.qz * qn);
    const V3<T> dl = mulT(R, mk<T>(ex.lin.x - lin0.x, ex.lin.y - lin0.y, ex.lin.z - lin0.z));
    const T sz = T(Const::sense_z), dax = ex.ang.x - ang0.x, day = ex.ang.y - ang0.y;
    acc = acc + dl + mk<T>(day * sz, -dax * sz, T(0));
  }
  integrate<T, LOAD>(M, s, im, c0, c1, c2, c3, h);
  return acc;
}

}  // namespace qd
