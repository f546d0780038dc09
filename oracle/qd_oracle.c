/*
 * qd_oracle.c -- CPU float64 restatement of the reference hot path.
 * TEST INFRASTRUCTURE ONLY (see qd_oracle.h for scope, citations and the
 * pinning status: physics step = "parity unpinned").
 */
#include "qd_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.14159265358979323846
#define MJMINVAL 1e-15

/* ------------------------------------------------------------------ utils */
static void v3cross(const double a[3], const double b[3], double o[3]) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static double v3dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void m3v(const double M[9], const double v[3], double o[3]) { /* o = M v */
  double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  double y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  double z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void m3tv(const double M[9], const double v[3], double o[3]) { /* o = M^T v */
  double x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  double y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  double z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void m3m(const double A[9], const double B[9], double O[9]) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(O, T, sizeof T);
}
static void m3t(const double A[9], double O[9]) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * j + i];
  memcpy(O, T, sizeof T);
}
static void rotx(double a, double R[9]) {
  double c = cos(a), s = sin(a);
  double T[9] = {1, 0, 0, 0, c, -s, 0, s, c};
  memcpy(R, T, sizeof T);
}
static void roty(double a, double R[9]) {
  double c = cos(a), s = sin(a);
  double T[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
  memcpy(R, T, sizeof T);
}
static void rotz(double a, double R[9]) {
  double c = cos(a), s = sin(a);
  double T[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
  memcpy(R, T, sizeof T);
}
static void quat_mul(const double a[4], const double b[4], double o[4]) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static void quat_norm(double q[4]) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MJMINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}

double orc_round5g(double x) {
  char buf[64];
  snprintf(buf, sizeof buf, "%.5g", x);
  return strtod(buf, NULL);
}

/* ------------------------------------------------ transformation.py:5-29 */
void orc_quat2dcm(const double qin[4], double R[9]) {
  /* scipy normalises the quaternion (transformation.py:13) */
  double q[4] = {qin[0], qin[1], qin[2], qin[3]};
  quat_norm(q);
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
void orc_quat2rpy(const double qin[4], double rpy[3]) {
  /* intrinsic 'ZYX' (R = Rz(yaw) Ry(pitch) Rx(roll)), reversed to roll,pitch,yaw
   * (transformation.py:16-18) */
  double q[4] = {qin[0], qin[1], qin[2], qin[3]};
  quat_norm(q);
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double sp = 2 * (w * y - z * x);
  if (sp > 1) sp = 1;
  if (sp < -1) sp = -1;
  rpy[0] = atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
  rpy[1] = asin(sp);
  rpy[2] = atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z));
}
void orc_rpy2quat(const double rpy[3], double q[4]) { /* transformation.py:21-24 */
  double cr = cos(rpy[0] / 2), sr = sin(rpy[0] / 2);
  double cp = cos(rpy[1] / 2), sp = sin(rpy[1] / 2);
  double cy = cos(rpy[2] / 2), sy = sin(rpy[2] / 2);
  q[0] = cr * cp * cy + sr * sp * sy;
  q[1] = sr * cp * cy - cr * sp * sy;
  q[2] = cr * sp * cy + sr * cp * sy;
  q[3] = cr * cp * sy - sr * sp * cy;
}
void orc_pendrp2quat(const double rp[2], double q[4]) { /* transformation.py:27-29: R = Rx(a) Ry(b) */
  double ca = cos(rp[0] / 2), sa = sin(rp[0] / 2), cb = cos(rp[1] / 2), sb = sin(rp[1] / 2);
  q[0] = ca * cb; q[1] = sa * cb; q[2] = ca * sb; q[3] = sa * sb;
}
void orc_dcm2quat(const double R[9], double q[4]) { /* transformation.py:5-8 (sign: scipy keeps as computed) */
  double tr = R[0] + R[4] + R[8];
  double w, x, y, z;
  /* scipy's from_matrix: pick the largest of (R00,R11,R22,trace) */
  double d[4] = {R[0], R[4], R[8], tr};
  int k = 0;
  for (int i = 1; i < 4; i++) if (d[i] > d[k]) k = i;
  if (k == 3) {
    x = R[7] - R[5]; y = R[2] - R[6]; z = R[3] - R[1]; w = 1 + tr;
  } else {
    int i = k, j = (i + 1) % 3, l = (j + 1) % 3;
    double v[4];
    v[i] = 1 - tr + 2 * R[4 * i];
    v[j] = R[3 * j + i] + R[3 * i + j];
    v[l] = R[3 * l + i] + R[3 * i + l];
    v[3] = R[3 * l + j] - R[3 * j + l];
    x = v[0]; y = v[1]; z = v[2]; w = v[3];
  }
  double n = sqrt(w * w + x * x + y * y + z * z);
  q[0] = w / n; q[1] = x / n; q[2] = y / n; q[3] = z / n;
}
static void rpy2dcm(const double rpy[3], double R[9]) {
  double q[4];
  orc_rpy2quat(rpy, q);
  orc_quat2dcm(q, R);
}

/* -------------------------------------------------- model (env_gen.py) */
/* symmetric 3x3 eigen-decomposition by quaternion Jacobi sweeps, eigenvalues
 * sorted in decreasing order (the convention MuJoCo's compiler uses for
 * body_inertia / body_iquat) */
static void eig3(const double A[9], double eval[3], double evec[9]) {
  double quat[4] = {1, 0, 0, 0}, D[9], tmp[9], V[9];
  const double eps = 1e-12;
  for (int iter = 0; iter < 500; iter++) {
    orc_quat2dcm(quat, V);
    double Vt[9];
    m3t(V, Vt);
    m3m(Vt, A, tmp);
    m3m(tmp, V, D);
    eval[0] = D[0]; eval[1] = D[4]; eval[2] = D[8];
    int rk, ck, rotk;
    if (fabs(D[1]) > fabs(D[2]) && fabs(D[1]) > fabs(D[5])) { rk = 0; ck = 1; rotk = 2; }
    else if (fabs(D[2]) > fabs(D[5])) { rk = 0; ck = 2; rotk = 1; }
    else { rk = 1; ck = 2; rotk = 0; }
    if (fabs(D[3 * rk + ck]) < eps) break;
    double tau = (D[4 * ck] - D[4 * rk]) / (2 * D[3 * rk + ck]);
    double t = tau >= 0 ? 1.0 / (tau + sqrt(1 + tau * tau)) : -1.0 / (-tau + sqrt(1 + tau * tau));
    double c = 1.0 / sqrt(1 + t * t);
    if (c > 1.0 - eps) break;
    double r[4] = {0, 0, 0, 0};
    r[rotk + 1] = tau >= 0 ? -sqrt(0.5 - 0.5 * c) : sqrt(0.5 - 0.5 * c);
    if (rotk == 1) r[rotk + 1] = -r[rotk + 1];
    r[0] = sqrt(1.0 - r[rotk + 1] * r[rotk + 1]);
    quat_norm(r);
    double nq[4];
    quat_mul(quat, r, nq);
    memcpy(quat, nq, sizeof nq);
    quat_norm(quat);
  }
  for (int j = 0; j < 3; j++) { /* bubble sort 0,1,0 into decreasing order */
    int j1 = j % 2;
    if (eval[j1] + eps < eval[j1 + 1]) {
      double t = eval[j1]; eval[j1] = eval[j1 + 1]; eval[j1 + 1] = t;
      double r[4] = {0.707106781186548, 0, 0, 0}, nq[4];
      r[(j1 + 2) % 3 + 1] = r[0];
      quat_mul(quat, r, nq);
      memcpy(quat, nq, sizeof nq);
      quat_norm(quat);
    }
  }
  orc_quat2dcm(quat, evec);
}

static void box_inertia(double m, double a, double b, double c, double I[3]) {
  I[0] = m / 3 * (b * b + c * c); I[1] = m / 3 * (a * a + c * c); I[2] = m / 3 * (a * a + b * b);
}
static void cyl_inertia(double m, double r, double hh, double I[3]) { /* axis z, half height hh */
  I[0] = I[1] = m * (3 * r * r + 4 * hh * hh) / 12; I[2] = m * r * r / 2;
}
/* add a geom (diag inertia Ig in its own frame rotated by Rg, mass m at pos)
 * to a running full inertia about `com` (xx,yy,zz,xy,xz,yz) */
static void add_geom(double tot[6], double m, const double Ig[3], const double Rg[9], const double pos[3],
                     const double com[3]) {
  double G[9] = {0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) G[3 * i + j] += Rg[3 * i + k] * Ig[k] * Rg[3 * j + k];
  double d[3] = {pos[0] - com[0], pos[1] - com[1], pos[2] - com[2]};
  double dd = v3dot(d, d);
  tot[0] += G[0] + m * (dd - d[0] * d[0]);
  tot[1] += G[4] + m * (dd - d[1] * d[1]);
  tot[2] += G[8] + m * (dd - d[2] * d[2]);
  tot[3] += G[1] - m * d[0] * d[1];
  tot[4] += G[2] - m * d[0] * d[2];
  tot[5] += G[5] - m * d[1] * d[2];
}
static void inertia_box_dims(const double I[3], double mass, double box[3]) {
  box[0] = sqrt(fmax(MJMINVAL, I[1] + I[2] - I[0]) / mass * 6.0);
  box[1] = sqrt(fmax(MJMINVAL, I[0] + I[2] - I[1]) / mass * 6.0);
  box[2] = sqrt(fmax(MJMINVAL, I[0] + I[1] - I[2]) / mass * 6.0);
}

static void floor_invweight(OrcModel *o);
void orc_build_model(const double raw[6], OrcModel *o) {
  memset(o, 0, sizeof *o);
  const double mass = raw[0], arm_len = raw[1], motor_force = raw[2], motor_tau = raw[3];
  const double pl = raw[4], wm = raw[5];
  const double hb = 0.05; /* env_gen.py:38 */
  o->gravity = 9.81;
  o->density = orc_round5g(1.2);       /* env_gen.py:83 */
  o->viscosity = orc_round5g(0.00002); /* env_gen.py:84 */
  o->damping = orc_round5g(0.15);      /* env_gen.py:23 */
  o->load = (pl > 0 && wm > 0);        /* env_gen.py:33-35 */

  /* --- core body geoms (env_gen.py:41-61) --- */
  const double body_mass = orc_round5g(0.56 * mass);
  const double arm_mass = orc_round5g(0.07 * mass);
  const double motor_mass = orc_round5g(0.04 * mass);
  double gm[9], gI[9][3], gR[9][9], gp[9][3];
  int ng = 0;
  const double Id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  /* core box */
  gm[ng] = body_mass;
  box_inertia(body_mass, orc_round5g(hb), orc_round5g(hb), orc_round5g(hb / 3), gI[ng]);
  memcpy(gR[ng], Id, sizeof Id);
  gp[ng][0] = gp[ng][1] = gp[ng][2] = 0;
  ng++;
  for (int i = 0; i < 4; i++) {
    double theta = i * PI / 2 - PI / 4;
    double A = sqrt(2.0) * hb + 0.5 * arm_len, B = sqrt(2.0) * hb + arm_len;
    /* arm box */
    gm[ng] = arm_mass;
    box_inertia(arm_mass, orc_round5g(arm_len / 2), orc_round5g(arm_len / 20), orc_round5g(arm_len / 20), gI[ng]);
    rotz(orc_round5g(theta), gR[ng]);
    gp[ng][0] = orc_round5g(A * cos(theta)); gp[ng][1] = orc_round5g(A * sin(theta)); gp[ng][2] = 0;
    ng++;
    /* motor cylinder */
    gm[ng] = motor_mass;
    cyl_inertia(motor_mass, orc_round5g(0.01), orc_round5g(0.01), gI[ng]);
    memcpy(gR[ng], Id, sizeof Id);
    gp[ng][0] = orc_round5g(B * cos(theta)); gp[ng][1] = orc_round5g(B * sin(theta)); gp[ng][2] = orc_round5g(0.015);
    ng++;
    /* motor site + actuator gear (env_gen.py:59,62-64) */
    o->rotor[i][0] = orc_round5g(B * cos(theta)); o->rotor[i][1] = orc_round5g(B * sin(theta)); o->rotor[i][2] = 0;
    o->gearT[i] = orc_round5g(motor_force / 100 * ((i % 2) ? -1.0 : 1.0));
  }
  o->gearF = orc_round5g(motor_force);
  o->tau = orc_round5g(motor_tau);
  o->sense[0] = 0; o->sense[1] = 0; o->sense[2] = orc_round5g(-hb / 4); /* env_gen.py:48 */

  /* MuJoCo compile, multi-geom body: total mass, COM, full inertia, principal axes */
  double com[3] = {0, 0, 0};
  o->m0 = 0;
  for (int g = 0; g < ng; g++) {
    o->m0 += gm[g];
    for (int k = 0; k < 3; k++) com[k] += gm[g] * gp[g][k];
  }
  for (int k = 0; k < 3; k++) o->c0[k] = com[k] / o->m0;
  double tot[6] = {0};
  for (int g = 0; g < ng; g++) add_geom(tot, gm[g], gI[g], gR[g], gp[g], o->c0);
  memcpy(o->I0full, tot, sizeof tot);
  double A[9] = {tot[0], tot[3], tot[4], tot[3], tot[1], tot[5], tot[4], tot[5], tot[2]};
  eig3(A, o->I0, o->R0i);
  inertia_box_dims(o->I0, o->m0, o->box0);

  if (o->load) {
    /* link (env_gen.py:66-68): single geom -> inertial frame = geom frame */
    o->anchor[0] = 0; o->anchor[1] = 0; o->anchor[2] = orc_round5g(-hb / 2);
    o->m1 = orc_round5g(0.01);
    double r = orc_round5g(0.02);
    o->I1 = 0.4 * o->m1 * r * r;
    double I1v[3] = {o->I1, o->I1, o->I1}, b1[3];
    inertia_box_dims(I1v, o->m1, b1);
    o->box1 = b1[0];
    /* pendulum (env_gen.py:69-72): rod cylinder + load box, both on the z axis */
    double pole_mass = orc_round5g(0.2 * pl);
    double wmass = orc_round5g(wm);
    double rod_r = orc_round5g(0.005), rod_hh = orc_round5g(pl / 2), rod_z = orc_round5g(-pl / 2);
    double bs = orc_round5g(0.1 * cbrt(wm)), box_z = orc_round5g(-pl);
    double Irod[3], Ibox[3];
    cyl_inertia(pole_mass, rod_r, rod_hh, Irod);
    box_inertia(wmass, bs, bs, bs, Ibox);
    o->m2 = pole_mass + wmass;
    double cz = (pole_mass * rod_z + wmass * box_z) / o->m2;
    o->lc = -cz;
    double c2[3] = {0, 0, cz}, t2[6] = {0};
    double p_rod[3] = {0, 0, rod_z}, p_box[3] = {0, 0, box_z};
    add_geom(t2, pole_mass, Irod, Id, p_rod, c2);
    add_geom(t2, wmass, Ibox, Id, p_box, c2);
    /* already diagonal with Ixx = Iyy >= Izz: principal frame = body frame */
    o->I2[0] = t2[0]; o->I2[1] = t2[1]; o->I2[2] = t2[2];
    inertia_box_dims(o->I2, o->m2, o->box2);
  }
  memcpy(o->raw, raw, sizeof o->raw);
  floor_invweight(o);
}

/* ---------------------------------------------------------- dynamics */
typedef struct {
  int nv, nb;
  double M[64], Q[8], bias[8]; /* M qacc = Q - bias */
  double R[9];                  /* base orientation */
  double omega0[3];             /* world angular velocity of base */
} Dyn;

/* inertia-box fluid forces on one body: wrench at its COM, world frame */
static void fluid_wrench(const OrcModel *m, const double Ri[9] /* inertial axes in world (columns) */,
                         const double box[3], const double omega_w[3], const double vcom_w[3], double f_w[3],
                         double t_w[3]) {
  double la[3], ll[3], fa[3] = {0, 0, 0}, fl[3] = {0, 0, 0};
  m3tv(Ri, omega_w, la);
  m3tv(Ri, vcom_w, ll);
  if (m->viscosity > 0) {
    double diam = (box[0] + box[1] + box[2]) / 3.0;
    for (int k = 0; k < 3; k++) {
      fa[k] = -PI * diam * diam * diam * m->viscosity * la[k];
      fl[k] = -3.0 * PI * diam * m->viscosity * ll[k];
    }
  }
  if (m->density > 0) {
    fl[0] -= 0.5 * m->density * box[1] * box[2] * fabs(ll[0]) * ll[0];
    fl[1] -= 0.5 * m->density * box[0] * box[2] * fabs(ll[1]) * ll[1];
    fl[2] -= 0.5 * m->density * box[0] * box[1] * fabs(ll[2]) * ll[2];
    fa[0] -= m->density * box[0] * (pow(box[1], 4) + pow(box[2], 4)) * fabs(la[0]) * la[0] / 64.0;
    fa[1] -= m->density * box[1] * (pow(box[0], 4) + pow(box[2], 4)) * fabs(la[1]) * la[1] / 64.0;
    fa[2] -= m->density * box[2] * (pow(box[0], 4) + pow(box[1], 4)) * fabs(la[2]) * la[2] / 64.0;
  }
  m3v(Ri, fl, f_w);
  m3v(Ri, fa, t_w);
}

/* World-frame projected Newton-Euler (Kane).  Generalized speeds follow
 * MuJoCo's free-joint convention: qvel[0:3] world-frame linear velocity of the
 * base frame origin, qvel[3:6] body-frame angular velocity, then hinge rates. */
static void dyn_terms(const OrcModel *m, const double *qpos, const double *qvel, const double act[4], Dyn *d) {
  const int load = m->load;
  const int nv = load ? 8 : 6, nb = load ? 3 : 1;
  d->nv = nv; d->nb = nb;
  double q[4] = {qpos[3], qpos[4], qpos[5], qpos[6]};
  quat_norm(q);
  double R[9];
  orc_quat2dcm(q, R);
  memcpy(d->R, R, sizeof R);
  const double *p = qpos;
  const double *vlin = qvel, *wb = qvel + 3;
  double th1 = load ? qpos[7] : 0, th2 = load ? qpos[8] : 0;
  double thd1 = load ? qvel[6] : 0, thd2 = load ? qvel[7] : 0;

  /* body poses */
  double Rb[3][9], xc[3][3], mass[3], Iw[3][9];
  double tmp[3], Rx[9], Ry[9];
  /* body 0 */
  memcpy(Rb[0], R, sizeof R);
  m3v(R, m->c0, tmp);
  for (int k = 0; k < 3; k++) xc[0][k] = p[k] + tmp[k];
  mass[0] = m->m0;
  {
    double I0[9] = {m->I0full[0], m->I0full[3], m->I0full[4], m->I0full[3], m->I0full[1],
                    m->I0full[5], m->I0full[4], m->I0full[5], m->I0full[2]};
    double Rt[9], T[9];
    m3t(R, Rt);
    m3m(R, I0, T);
    m3m(T, Rt, Iw[0]);
  }
  double xa[3] = {0, 0, 0}, axis1[3] = {0, 0, 0}, axis2[3] = {0, 0, 0};
  if (load) {
    m3v(R, m->anchor, tmp);
    for (int k = 0; k < 3; k++) xa[k] = p[k] + tmp[k];
    rotx(th1, Rx);
    roty(th2, Ry);
    m3m(R, Rx, Rb[1]);
    m3m(Rb[1], Ry, Rb[2]);
    for (int k = 0; k < 3; k++) { axis1[k] = R[3 * k + 0]; axis2[k] = Rb[1][3 * k + 1]; }
    for (int k = 0; k < 3; k++) xc[1][k] = xa[k];
    mass[1] = m->m1;
    for (int k = 0; k < 9; k++) Iw[1][k] = 0;
    Iw[1][0] = Iw[1][4] = Iw[1][8] = m->I1;
    double down[3] = {0, 0, -m->lc};
    m3v(Rb[2], down, tmp);
    for (int k = 0; k < 3; k++) xc[2][k] = xa[k] + tmp[k];
    mass[2] = m->m2;
    double I2[9] = {m->I2[0], 0, 0, 0, m->I2[1], 0, 0, 0, m->I2[2]}, Rt[9], T[9];
    m3t(Rb[2], Rt);
    m3m(Rb[2], I2, T);
    m3m(T, Rt, Iw[2]);
  }

  /* Jacobian columns */
  double Jv[3][8][3], Jw[3][8][3];
  memset(Jv, 0, sizeof Jv);
  memset(Jw, 0, sizeof Jw);
  for (int b = 0; b < nb; b++) {
    for (int j = 0; j < 3; j++) Jv[b][j][j] = 1.0;
    for (int j = 0; j < 3; j++) {
      double a[3] = {R[3 * 0 + j], R[3 * 1 + j], R[3 * 2 + j]};
      double r[3] = {xc[b][0] - p[0], xc[b][1] - p[1], xc[b][2] - p[2]};
      memcpy(Jw[b][3 + j], a, sizeof a);
      v3cross(a, r, Jv[b][3 + j]);
    }
    if (load && b >= 1) {
      double r[3] = {xc[b][0] - xa[0], xc[b][1] - xa[1], xc[b][2] - xa[2]};
      memcpy(Jw[b][6], axis1, sizeof axis1);
      v3cross(axis1, r, Jv[b][6]);
      if (b == 2) {
        memcpy(Jw[b][7], axis2, sizeof axis2);
        v3cross(axis2, r, Jv[b][7]);
      }
    }
  }

  /* velocities and velocity-product accelerations */
  double w[3][3], vc[3][3], al[3][3], ac[3][3];
  m3v(R, wb, w[0]);
  memcpy(d->omega0, w[0], sizeof w[0]);
  {
    double r[3] = {xc[0][0] - p[0], xc[0][1] - p[1], xc[0][2] - p[2]}, t[3];
    v3cross(w[0], r, t);
    for (int k = 0; k < 3; k++) vc[0][k] = vlin[k] + t[k];
    v3cross(w[0], t, ac[0]);
    al[0][0] = al[0][1] = al[0][2] = 0;
  }
  if (load) {
    double ra[3] = {xa[0] - p[0], xa[1] - p[1], xa[2] - p[2]}, t[3], va[3], aa[3];
    v3cross(w[0], ra, t);
    for (int k = 0; k < 3; k++) va[k] = vlin[k] + t[k];
    v3cross(w[0], t, aa);
    /* body 1 */
    double j1[3] = {thd1 * axis1[0], thd1 * axis1[1], thd1 * axis1[2]};
    for (int k = 0; k < 3; k++) w[1][k] = w[0][k] + j1[k];
    v3cross(w[0], j1, al[1]);
    memcpy(vc[1], va, sizeof va);
    memcpy(ac[1], aa, sizeof aa);
    /* body 2 */
    double j2[3] = {thd2 * axis2[0], thd2 * axis2[1], thd2 * axis2[2]};
    for (int k = 0; k < 3; k++) w[2][k] = w[1][k] + j2[k];
    v3cross(w[1], j2, t);
    for (int k = 0; k < 3; k++) al[2][k] = al[1][k] + t[k];
    double r[3] = {xc[2][0] - xa[0], xc[2][1] - xa[1], xc[2][2] - xa[2]}, t2[3], t3[3];
    v3cross(w[2], r, t);
    for (int k = 0; k < 3; k++) vc[2][k] = va[k] + t[k];
    v3cross(w[2], t, t2);
    v3cross(al[2], r, t3);
    for (int k = 0; k < 3; k++) ac[2][k] = aa[k] + t3[k] + t2[k];
  }

  /* M, bias (incl. gravity) */
  memset(d->M, 0, sizeof d->M);
  memset(d->bias, 0, sizeof d->bias);
  memset(d->Q, 0, sizeof d->Q);
  for (int b = 0; b < nb; b++) {
    double Iwv[3], gy[3], N[3], F[3];
    m3v(Iw[b], w[b], Iwv);
    v3cross(w[b], Iwv, gy);
    m3v(Iw[b], al[b], N);
    for (int k = 0; k < 3; k++) { N[k] += gy[k]; F[k] = mass[b] * ac[b][k]; }
    F[2] += mass[b] * m->gravity; /* - m g, g = (0,0,-9.81) */
    for (int i = 0; i < nv; i++) {
      d->bias[i] += v3dot(Jv[b][i], F) + v3dot(Jw[b][i], N);
      double IJ[3];
      m3v(Iw[b], Jw[b][i], IJ);
      for (int j = 0; j < nv; j++)
        d->M[8 * i + j] += mass[b] * v3dot(Jv[b][j], Jv[b][i]) + v3dot(Jw[b][j], IJ);
    }
  }

  /* passive: hinge damping + fluid */
  if (load) {
    d->Q[6] -= m->damping * thd1;
    d->Q[7] -= m->damping * thd2;
  }
  for (int b = 0; b < nb; b++) {
    double Ri[9], box[3], f[3], t[3];
    if (b == 0) { m3m(R, m->R0i, Ri); memcpy(box, m->box0, sizeof box); }
    else if (b == 1) { memcpy(Ri, Rb[1], sizeof Ri); box[0] = box[1] = box[2] = m->box1; }
    else { memcpy(Ri, Rb[2], sizeof Ri); memcpy(box, m->box2, sizeof box); }
    fluid_wrench(m, Ri, box, w[b], vc[b], f, t);
    for (int i = 0; i < nv; i++) d->Q[i] += v3dot(Jv[b][i], f) + v3dot(Jw[b][i], t);
  }
  /* actuation: site transmission, gear (0,0,F,0,0,T) in the site frame */
  for (int r = 0; r < 4; r++) {
    double fl[3] = {0, 0, m->gearF * act[r]}, tl[3] = {0, 0, m->gearT[r] * act[r]}, fw[3], tw[3], rs[3];
    m3v(R, fl, fw);
    m3v(R, tl, tw);
    m3v(R, m->rotor[r], rs);
    for (int j = 0; j < 3; j++) d->Q[j] += fw[j];
    for (int j = 0; j < 3; j++) {
      double a[3] = {R[3 * 0 + j], R[3 * 1 + j], R[3 * 2 + j]}, jv[3];
      v3cross(a, rs, jv);
      d->Q[3 + j] += v3dot(jv, fw) + v3dot(a, tw);
    }
  }
}

/* dense symmetric positive-definite solve (Cholesky), n <= 8, row stride 8 */
static void spd_solve(const double *Min, int n, const double *b, double *x) {
  double L[64];
  memset(L, 0, sizeof L);
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = Min[8 * i + j];
      for (int k = 0; k < j; k++) s -= L[8 * i + k] * L[8 * j + k];
      L[8 * i + j] = (i == j) ? sqrt(s) : s / L[8 * j + j];
    }
  double y[8];
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[8 * i + k] * y[k];
    y[i] = s / L[8 * i + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= L[8 * k + i] * x[k];
    x[i] = s / L[8 * i + i];
  }
}

static void accel_sensor(const OrcModel *m, const Dyn *d, const double *qacc, double sensor[3]) {
  /* classical acceleration of the site minus gravity, in the site frame */
  double alpha_w[3], rs[3], t[3], t2[3], a[3];
  m3v(d->R, qacc + 3, alpha_w);
  m3v(d->R, m->sense, rs);
  v3cross(alpha_w, rs, t);
  v3cross(d->omega0, rs, t2);
  v3cross(d->omega0, t2, t2);
  for (int k = 0; k < 3; k++) a[k] = qacc[k] + t[k] + t2[k];
  a[2] += m->gravity;
  m3tv(d->R, a, sensor);
}

static double clamp01(double x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }

void orc_forward(const OrcModel *m, const double *qpos, const double *qvel, const double act[4],
                 const double ctrl[4], double *qacc, double act_dot[4], double sensor[3]) {
  Dyn d;
  dyn_terms(m, qpos, qvel, act, &d);
  double rhs[8];
  for (int i = 0; i < d.nv; i++) rhs[i] = d.Q[i] - d.bias[i];
  spd_solve(d.M, d.nv, rhs, qacc);
  for (int r = 0; r < 4; r++) act_dot[r] = (clamp01(ctrl[r]) - act[r]) / fmax(m->tau, MJMINVAL);
  accel_sensor(m, &d, qacc, sensor);
}

void orc_mass_matrix(const OrcModel *m, const double *qpos, double *M) {
  double qvel[8] = {0}, act[4] = {0};
  Dyn d;
  dyn_terms(m, qpos, qvel, act, &d);
  for (int i = 0; i < d.nv; i++)
    for (int j = 0; j < d.nv; j++) M[d.nv * i + j] = d.M[8 * i + j];
}

void orc_step(const OrcModel *m, double h, int nstep, double *qpos, double *qvel, double act[4],
              const double ctrl[4], double sensor[3]) {
  for (int s = 0; s < nstep; s++) {
    Dyn d;
    dyn_terms(m, qpos, qvel, act, &d);
    const int nv = d.nv;
    double rhs[8], qacc[8], qimp[8], act_dot[4];
    for (int i = 0; i < nv; i++) rhs[i] = d.Q[i] - d.bias[i];
    spd_solve(d.M, nv, rhs, qacc);
    accel_sensor(m, &d, qacc, sensor);
    for (int r = 0; r < 4; r++) act_dot[r] = (clamp01(ctrl[r]) - act[r]) / fmax(m->tau, MJMINVAL);
    /* Euler with implicit joint damping: (M + h diag(damping)) a = M qacc */
    if (m->load) {
      double Mh[64];
      memcpy(Mh, d.M, sizeof Mh);
      Mh[8 * 6 + 6] += h * m->damping;
      Mh[8 * 7 + 7] += h * m->damping;
      spd_solve(Mh, nv, rhs, qimp);
    } else {
      memcpy(qimp, qacc, sizeof qimp);
    }
    for (int r = 0; r < 4; r++) act[r] += h * act_dot[r];
    for (int i = 0; i < nv; i++) qvel[i] += h * qimp[i];
    for (int k = 0; k < 3; k++) qpos[k] += h * qvel[k];
    {
      double q[4] = {qpos[3], qpos[4], qpos[5], qpos[6]};
      quat_norm(q);
      double wn = sqrt(qvel[3] * qvel[3] + qvel[4] * qvel[4] + qvel[5] * qvel[5]);
      double ax[3] = {1, 0, 0};
      if (wn >= MJMINVAL) { ax[0] = qvel[3] / wn; ax[1] = qvel[4] / wn; ax[2] = qvel[5] / wn; } else wn = 0;
      double ang = h * wn, sn = sin(ang / 2);
      double qr[4] = {cos(ang / 2), ax[0] * sn, ax[1] * sn, ax[2] * sn}, qn[4];
      quat_mul(q, qr, qn);
      quat_norm(qn);
      for (int k = 0; k < 4; k++) qpos[3 + k] = qn[k];
    }
    if (m->load) { qpos[7] += h * qvel[6]; qpos[8] += h * qvel[7]; }
  }
}

void orc_energy_momentum(const OrcModel *m, const double *qpos, const double *qvel, double *kinetic,
                         double *potential, double lin[3], double ang[3]) {
  /* recompute kinematics (same as dyn_terms) via M and Jacobians: KE = 1/2 v^T M v */
  double act[4] = {0};
  Dyn d;
  dyn_terms(m, qpos, qvel, act, &d);
  double ke = 0;
  for (int i = 0; i < d.nv; i++)
    for (int j = 0; j < d.nv; j++) ke += 0.5 * qvel[i] * d.M[8 * i + j] * qvel[j];
  *kinetic = ke;
  /* COM positions, velocities */
  double R[9];
  memcpy(R, d.R, sizeof R);
  double x[3][3], v[3][3], w[3][3], mass[3], Iw[3][9], tmp[3];
  int nb = d.nb;
  const double *p = qpos;
  m3v(R, m->c0, tmp);
  for (int k = 0; k < 3; k++) x[0][k] = p[k] + tmp[k];
  m3v(R, qvel + 3, w[0]);
  v3cross(w[0], tmp, v[0]);
  for (int k = 0; k < 3; k++) v[0][k] += qvel[k];
  mass[0] = m->m0;
  {
    double I0[9] = {m->I0full[0], m->I0full[3], m->I0full[4], m->I0full[3], m->I0full[1],
                    m->I0full[5], m->I0full[4], m->I0full[5], m->I0full[2]}, Rt[9], T[9];
    m3t(R, Rt); m3m(R, I0, T); m3m(T, Rt, Iw[0]);
  }
  if (m->load) {
    double Rx[9], Ry[9], R1[9], R2[9], xa[3], va[3], ra[3];
    rotx(qpos[7], Rx); roty(qpos[8], Ry);
    m3m(R, Rx, R1); m3m(R1, Ry, R2);
    m3v(R, m->anchor, ra);
    for (int k = 0; k < 3; k++) xa[k] = p[k] + ra[k];
    v3cross(w[0], ra, va);
    for (int k = 0; k < 3; k++) va[k] += qvel[k];
    for (int k = 0; k < 3; k++) { x[1][k] = xa[k]; v[1][k] = va[k]; w[1][k] = w[0][k] + qvel[6] * R[3 * k]; }
    mass[1] = m->m1;
    memset(Iw[1], 0, sizeof Iw[1]);
    Iw[1][0] = Iw[1][4] = Iw[1][8] = m->I1;
    double down[3] = {0, 0, -m->lc}, r[3], t[3];
    m3v(R2, down, r);
    for (int k = 0; k < 3; k++) { x[2][k] = xa[k] + r[k]; w[2][k] = w[1][k] + qvel[7] * R1[3 * k + 1]; }
    v3cross(w[2], r, t);
    for (int k = 0; k < 3; k++) v[2][k] = va[k] + t[k];
    mass[2] = m->m2;
    double I2[9] = {m->I2[0], 0, 0, 0, m->I2[1], 0, 0, 0, m->I2[2]}, Rt[9], T[9];
    m3t(R2, Rt); m3m(R2, I2, T); m3m(T, Rt, Iw[2]);
  }
  double mt = 0, xcom[3] = {0, 0, 0};
  *potential = 0;
  lin[0] = lin[1] = lin[2] = 0;
  for (int b = 0; b < nb; b++) {
    mt += mass[b];
    *potential += mass[b] * m->gravity * x[b][2];
    for (int k = 0; k < 3; k++) { xcom[k] += mass[b] * x[b][k]; lin[k] += mass[b] * v[b][k]; }
  }
  for (int k = 0; k < 3; k++) xcom[k] /= mt;
  ang[0] = ang[1] = ang[2] = 0;
  for (int b = 0; b < nb; b++) {
    double Iwv[3], r[3] = {x[b][0] - xcom[0], x[b][1] - xcom[1], x[b][2] - xcom[2]}, mv[3], t[3];
    m3v(Iw[b], w[b], Iwv);
    for (int k = 0; k < 3; k++) mv[k] = mass[b] * v[b][k];
    v3cross(r, mv, t);
    for (int k = 0; k < 3; k++) ang[k] += Iwv[k] + t[k];
  }
}

/* ---------------------------------------- state vector (BaseDroneEnv.py:357-380) */
int orc_drone_state(int load, const double *qpos, const double *qvel, const double sensor[3],
                    const double act[4], const double ref[4], const double raw[6], double *o) {
  int n = 0;
  double rpy[3];
  orc_quat2rpy(qpos + 3, rpy);
  for (int k = 0; k < 3; k++) o[n++] = qpos[k];
  for (int k = 0; k < 3; k++) o[n++] = rpy[k];
  for (int k = 0; k < 3; k++) o[n++] = qvel[k];
  for (int k = 0; k < 3; k++) o[n++] = qvel[3 + k];
  if (load) {
    o[n++] = qpos[7]; o[n++] = qpos[8];
    o[n++] = qvel[6]; o[n++] = qvel[7];
  }
  for (int k = 0; k < 3; k++) o[n++] = sensor[k];
  for (int k = 0; k < 4; k++) o[n++] = act[k];
  for (int k = 0; k < 4; k++) o[n++] = ref[k];
  for (int k = 0; k < 6; k++) o[n++] = raw[k];
  return n;
}

/* numpy float remainder (sign follows the divisor) */
static double npmod(double a, double b) {
  double m = fmod(a, b);
  if (m != 0) { if ((b < 0) != (m < 0)) m += b; } else m = copysign(0.0, b);
  return m;
}

enum {
  OBS_RAW = 0, OBS_GLOBAL_RPY, OBS_LOCAL_PRY, OBS_FULLSTATE, OBS_FULLSTATE_ZVEC, OBS_PRY_ACC,
  OBS_PRY_PARAMS, OBS_PRY_ACC_PARAMS, OBS_RPY_PARAMS, OBS_RPY_FAKEPARAMS, OBS_LOCAL_RPY,
  OBS_PRY_ACC_NOPEND, OBS_PRY_ACC_PARAMS_NOPEND, OBS_RM_PARAMS, OBS_ZVEC, OBS_SIMPLE
};

int orc_obs_dim(int kind, int ns) {
  int np = ns - 27;
  switch (kind) {
    case OBS_RAW: return ns;
    case OBS_GLOBAL_RPY: case OBS_LOCAL_PRY: case OBS_LOCAL_RPY: return 16;
    case OBS_FULLSTATE: return 23;
    case OBS_FULLSTATE_ZVEC: return 24;
    case OBS_PRY_ACC: return 19;
    case OBS_PRY_PARAMS: case OBS_RPY_PARAMS: return 16 + np;
    case OBS_RPY_FAKEPARAMS: return 22;
    case OBS_PRY_ACC_PARAMS: return 19 + np;
    case OBS_PRY_ACC_NOPEND: return 15;
    case OBS_PRY_ACC_PARAMS_NOPEND: return -1; /* NameError in the reference (observation_wrappers.py:448) */
    case OBS_RM_PARAMS: return 22 + np;
    case OBS_ZVEC: return 17;
    case OBS_SIMPLE: return 6;
  }
  return -1;
}

int orc_obs(int kind, const double *s, int ns, const double ref[4], double *o) {
  const int np = ns - 27;
  const double *par = s + 27;
  if (kind == OBS_RAW) { memcpy(o, s, ns * sizeof(double)); return ns; }
  if (kind == OBS_PRY_ACC_PARAMS_NOPEND) return -1;
  const double *xyz = s, *rpy = s + 3, *vel = s + 6, *angv = s + 9, *prp = s + 12, *pw = s + 14;
  const double *acc = s + 16, *act = s + 19;
  double hd = npmod(ref[3] - rpy[2] + PI, 2 * PI) - PI;
  double eg[3] = {ref[0] - xyz[0], ref[1] - xyz[1], ref[2] - xyz[2]};
  double R[9], el[3], vl[3];
  rpy2dcm(rpy, R);
  m3tv(R, eg, el);
  m3tv(R, vel, vl);
  double rp0[3] = {rpy[0], rpy[1], 0}, Rz0[9];
  rpy2dcm(rp0, Rz0);
  double zvec[3] = {Rz0[2], Rz0[5], Rz0[8]};
  int n = 0;
#define PUT3(v) do { o[n++] = (v)[0]; o[n++] = (v)[1]; o[n++] = (v)[2]; } while (0)
#define PUT2(v) do { o[n++] = (v)[0]; o[n++] = (v)[1]; } while (0)
#define PUT2R(v) do { o[n++] = (v)[1]; o[n++] = (v)[0]; } while (0)
#define PUTP() do { for (int k = 0; k < np; k++) o[n++] = par[k]; } while (0)
  switch (kind) {
    case OBS_GLOBAL_RPY: PUT3(eg); PUT2(rpy); o[n++] = hd; PUT3(vel); PUT3(angv); PUT2(prp); PUT2(pw); break;
    case OBS_LOCAL_PRY: PUT3(el); PUT2R(rpy); o[n++] = hd; PUT3(vl); PUT3(angv); PUT2R(prp); PUT2(pw); break;
    case OBS_FULLSTATE:
      PUT3(el); PUT2R(rpy); o[n++] = hd; PUT3(vl); PUT3(angv); PUT3(acc);
      for (int k = 0; k < 4; k++) o[n++] = act[k];
      PUT2R(prp); PUT2(pw); break;
    case OBS_FULLSTATE_ZVEC:
      PUT3(el); PUT3(zvec); o[n++] = hd; PUT3(vl); PUT3(angv); PUT3(acc);
      for (int k = 0; k < 4; k++) o[n++] = act[k];
      PUT2R(prp); PUT2(pw); break;
    case OBS_PRY_ACC: PUT3(el); PUT2R(rpy); o[n++] = hd; PUT3(vl); PUT3(angv); PUT3(acc); PUT2R(prp); PUT2(pw); break;
    case OBS_PRY_PARAMS: PUT3(el); PUT2R(rpy); o[n++] = hd; PUT3(vl); PUT3(angv); PUT2R(prp); PUT2(pw); PUTP(); break;
    case OBS_PRY_ACC_PARAMS:
      PUT3(el); PUT2R(rpy); o[n++] = hd; PUT3(vl); PUT3(angv); PUT2R(prp); PUT3(acc); PUT2(pw); PUTP(); break;
    case OBS_RPY_PARAMS: PUT3(el); PUT2(rpy); o[n++] = hd; PUT3(vl); PUT3(angv); PUT2(prp); PUT2(pw); PUTP(); break;
    case OBS_RPY_FAKEPARAMS: {
      const double fake[6] = {1, 0.17, 7, 0.01, 1.2, 0.3};
      PUT3(el); PUT2(rpy); o[n++] = hd; PUT3(vl); PUT3(angv); PUT2(prp); PUT2(pw);
      for (int k = 0; k < 6; k++) o[n++] = fake[k];
      break;
    }
    case OBS_LOCAL_RPY: PUT3(el); PUT2(rpy); o[n++] = hd; PUT3(vl); PUT3(angv); PUT2(prp); PUT2(pw); break;
    case OBS_PRY_ACC_NOPEND: PUT3(el); PUT2R(rpy); o[n++] = hd; PUT3(vl); PUT3(angv); PUT3(acc); break;
    case OBS_RM_PARAMS: {
      double r3[3] = {rpy[0], rpy[1], -hd}, Rm[9], RmT[9];
      rpy2dcm(r3, Rm);
      m3t(Rm, RmT);
      PUT3(el);
      for (int k = 0; k < 9; k++) o[n++] = RmT[k];
      PUT3(vl); PUT3(angv); PUT2(prp); PUT2(pw); PUTP(); break;
    }
    case OBS_ZVEC: PUT3(el); PUT3(zvec); o[n++] = hd; PUT3(vl); PUT3(angv); PUT2(prp); PUT2(pw); break;
    default: return -1;
  }
  return n;
}

void orc_simple_obs(const double qpos[7], double o[6]) {
  /* SimpleDrone.py:94-98: scipy is given MuJoCo's (w,x,y,z) as if it were
   * (x,y,z,w), then extrinsic 'zyx' Euler angles (R = Rx(a) Ry(b) Rz(c),
   * returned as [c, b, a]) */
  double q[4] = {qpos[6], qpos[3], qpos[4], qpos[5]}; /* (w',x',y',z') = (z, w, x, y) */
  double R[9];
  orc_quat2dcm(q, R);
  double sb = R[2];
  if (sb > 1) sb = 1;
  if (sb < -1) sb = -1;
  o[0] = qpos[0]; o[1] = qpos[1]; o[2] = qpos[2];
  o[3] = atan2(-R[1], R[0]);
  o[4] = asin(sb);
  o[5] = atan2(-R[5], R[8]);
}

/* ---------------------------------------------------- rewards.py:5-368 */
enum {
  REW_DEFAULT = 0, REW_DISTANCE, REW_DISTANCE_ENERGY, REW_PEND_ANGLE, REW_PEND_ANGLE2, REW_PEND_ANGLE3,
  REW_PEND_EN, REW_PEND_EN2, REW_PEND_EN3, REW_PEND_EN4, REW_DISTANCE_TIME_ENERGY, REW_REWARD_1,
  REW_PEND_DIST, REW_PEND_DIST_HEADING, REW_REWARD_2, REW_REWARD_2_PENERGY, REW_REWARD_3, REW_SIMPLE
};

static double sq3(const double *v) { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; }
static double sq2(const double *v) { return v[0] * v[0] + v[1] * v[1]; }

/* pendulum tip velocity^2, kinetic + potential pieces (rewards.py:82-104,177-184) */
static void pend_energy(const double *s, double *E, double *ph) {
  const double *par = s + 27, *prp = s + 12, *rpy = s + 3, *orp = s + 14, *om = s + 9;
  double Rd[9], Rp[9], qp[4], Rx[9], Ry[9];
  rpy2dcm(rpy, Rd);
  orc_pendrp2quat(prp, qp);
  orc_quat2dcm(qp, Rp);
  rotx(prp[0], Rx);
  roty(prp[1], Ry);
  double end[3] = {0, 0, -par[4]};
  double ox[9] = {0, 0, 0, 0, 0, -orp[0], 0, orp[0], 0};
  double oy[9] = {0, 0, orp[1], 0, 0, 0, -orp[1], 0, 0};
  double oc[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double t1[3], t2[3], T[9], U[9], v[3];
  m3v(Rp, end, t1); m3v(oc, t1, t1); m3v(Rd, t1, t1);           /* Rd oc Rp end */
  m3m(Rx, ox, T); m3m(T, Ry, T);                                  /* Rx ox Ry */
  m3m(Rx, Ry, U); m3m(U, oy, U);                                  /* Rx Ry oy */
  for (int k = 0; k < 9; k++) T[k] += U[k];
  m3v(T, end, t2); m3v(Rd, t2, t2);
  /* rewards.py:103-104: state[6:9] has shape (3,), the other two terms shape
   * (3,1), so numpy broadcasts the sum to a 3x3 matrix M[i][j] = vel[j] + c[i]
   * and the "energy" is the sum of all nine squares. */
  (void)v;
  double e = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { double x = s[6 + j] + (t1[i] + t2[i]); e += x * x; }
  *E = e;
  double r[3];
  m3v(Rp, end, r); m3v(Rd, r, r);
  *ph = r[2];
}
/* tether tip position with the ZYX pendulum rotation (rewards.py:285-292) */
static void pend_tip_zyx(const double *s, double len, double tip[3], double Rd[9], double Rp[9]) {
  double r3[3] = {s[12], s[13], 0}, end[3] = {0, 0, -len}, t[3];
  rpy2dcm(s + 3, Rd);
  rpy2dcm(r3, Rp);
  m3v(Rp, end, t); m3v(Rd, t, t);
  for (int k = 0; k < 3; k++) tip[k] = s[k] + t[k];
}

double orc_reward(int kind, const double *s, int ns, const double a[4], long k, const double ref[4],
                  double max_distance) {
  (void)ns;
  double dv[3] = {s[0] - ref[0], s[1] - ref[1], s[2] - ref[2]};
  double d2 = sq3(dv), d = sqrt(d2);
  double hraw = fabs(s[5] - ref[3]);
  double hw = npmod(hraw + PI, 2 * PI) - PI;
  double h1 = fabs(hw), h2 = hw * hw;
  double u2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
  double ad = sqrt(sq3(s + 3));
  double E, ph;
  switch (kind) {
    case REW_DEFAULT: return 3 - d;
    case REW_DISTANCE: return 5 - d - 0.1 * h1;
    case REW_DISTANCE_ENERGY: return 3.5 - d2 - 0.1 * h1 - 0.2 * u2;
    case REW_PEND_ANGLE: return 3.5 - d2 - 0.2 * h2 - 0.2 * u2 - 0.2 * sq2(s + 12);
    case REW_PEND_ANGLE2: return 3.5 - d2 - 0.5 * h2 - 0.4 * u2 - 0.2 * sq2(s + 12) - 0.1 * sq3(s + 9);
    case REW_PEND_ANGLE3: {
      double r = 3.5 - d2 - 0.5 * h2 - 0.4 * u2;
      r -= (0.1 * sq2(s + 12) + 0.2 * sq2(s + 14) - 0.3 * sq2(s + 3) - 0.4 * sq3(s + 9)) / (1 + 100 * d2);
      return r;
    }
    case REW_PEND_EN: pend_energy(s, &E, &ph); return 3.5 - d2 - 0.5 * h2 - 0.4 * u2 - 0.2 * E;
    case REW_PEND_EN2: {
      double c = 0;
      for (int i = 0; i < 4; i++) { double m = fmax(a[i] - 0.5, 0); c += m * m; }
      pend_energy(s, &E, &ph);
      double r = 3.5 - 2 * d - 0.6 * h2 - 0.6 * c;
      if (d < 0.15) r = r + 3 - 0.2 * E - 0.2 * ad;
      return r;
    }
    case REW_PEND_EN3: {
      double c = 0;
      for (int i = 0; i < 4; i++) { double m = fmax(a[i] - 0.5, 0); c += m * m; }
      pend_energy(s, &E, &ph);
      double tot = 0.5 * E + 9.81 * ph;
      return 7 - d - 0.4 * h2 - 0.1 * c - 0.1 * tot - 0.05 * ad;
    }
    case REW_PEND_EN4: {
      double c = 0;
      for (int i = 0; i < 4; i++) { double m = fmax(a[i] - 0.6, 0); c += m * m; }
      pend_energy(s, &E, &ph);
      double tot = 0.5 * E + 9.81 * ph;
      return 5 - d - 0.6 * h2 - 0.1 * c - (0.2 * tot + 0.05 * ad) / (0.5 + d);
    }
    case REW_DISTANCE_TIME_ENERGY: {
      double too_far = d2 > max_distance * max_distance ? 1.0 : 0.0;
      long kk = k >= 0 ? k / 50 : -((-k + 49) / 50);
      return -(1 + (double)kk) * d2 - 500 * too_far - h1 - 0.02 * u2;
    }
    case REW_REWARD_1: {
      double close = d2 < 0.2 ? 1.0 : 0.0, too_far = d2 > max_distance * max_distance - 3 ? 1.0 : 0.0;
      return (7 + 20 * close - 3 * d2 * (1 + (double)k / 150) - 10 * too_far - 0.3 * sq2(s + 3) - 0.7 * h2 -
              0.3 * u2 - 0.3 * sq3(s + 6) - 0.5 * sq2(s + 14)) / 10;
    }
    case REW_PEND_DIST: {
      double tip[3], Rd[9], Rp[9];
      pend_tip_zyx(s, s[27 + 5], tip, Rd, Rp);
      double e[3] = {tip[0] - ref[0], tip[1] - ref[1], tip[2] - ref[2]};
      return -sq3(e);
    }
    case REW_PEND_DIST_HEADING: case REW_REWARD_2: case REW_REWARD_2_PENERGY: case REW_REWARD_3: {
      double tip[3], Rd[9], Rp[9];
      pend_tip_zyx(s, s[27 + 4], tip, Rd, Rp);
      double e[3] = {tip[0] - ref[0], tip[1] - ref[1], tip[2] - ref[2]};
      double dp2 = sq3(e);
      if (kind == REW_PEND_DIST_HEADING) return 3 - dp2 - 0.1 * h1;
      if (kind == REW_REWARD_2) return 4 - dp2 - 0.001 * (double)k * dp2 - 0.1 * h1 - 0.05 * u2;
      double pom[3] = {s[14], s[15], 0}, end[3] = {0, 0, -s[27 + 4]}, r[3], vl[3], vg[3];
      m3v(Rp, end, r);
      v3cross(pom, r, vl);
      m3v(Rd, vl, vg);
      for (int i = 0; i < 3; i++) vg[i] += s[6 + i];
      double Ep = sq3(vg);
      if (kind == REW_REWARD_2_PENERGY)
        return 4 - dp2 - 0.2 * h1 - 0.006 * (double)k * (dp2 + 0.2 * h1) - 0.05 * u2 - 0.1 * Ep;
      double c = 0;
      for (int i = 0; i < 4; i++) { double m = fmin(a[i] - 0.5, 0); c += m * m; }
      return 4 - d2 - 0.2 * h1 - 0.006 * (double)k * (d2 + 0.2 * h1 + 0.01 * Ep) - 0.1 * c - 0.1 * Ep;
    }
    case REW_SIMPLE: return 0.1 - d; /* SimpleDrone.py:60 */
  }
  return 0.0 / 0.0;
}

int orc_truncated(const double *s, const double ref[4], long num_steps, double max_distance, long max_steps) {
  double dv[3] = {s[0] - ref[0], s[1] - ref[1], s[2] - ref[2]};
  return sqrt(sq3(dv)) > max_distance || num_steps >= max_steps;
}

/* ------------------------------------------------------------- Philox */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static float u32_to_unit(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

void orc_sample_draws_philox(uint64_t seed, uint32_t env, uint32_t episode, float z[15], float u[2]) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, w[20];
  for (uint32_t b = 0; b < 5; b++) {
    uint32_t ctr[4] = {env, episode, b, 1u};
    orc_philox4x32(ctr, key, w + 4 * b);
  }
  for (int i = 0; i < 8; i++) {
    float u1 = u32_to_unit(w[2 * i]), u2 = u32_to_unit(w[2 * i + 1]);
    float r = sqrtf(-2.0f * logf(u1)), a = 6.28318530717958647692f * u2;
    float z0 = r * cosf(a), z1 = r * sinf(a);
    if (2 * i < 15) z[2 * i] = z0;
    if (2 * i + 1 < 15) z[2 * i + 1] = z1;
  }
  u[0] = u32_to_unit(w[16]);
  u[1] = u32_to_unit(w[17]);
}

static double clipd(double x, double lim) { return x < -lim ? -lim : (x > lim ? lim : x); }

void orc_sample_state_from_draws(const OrcSampleCfg *c, const double z[15], const double u[2], double *qpos,
                                 double *qvel) {
  if (c->random_start) {
    double n = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    double r = c->max_pos_offset * cbrt(u[0]);
    for (int k = 0; k < 3; k++) qpos[k] = c->start_pos[k] + r * (z[k] / n);
    double rpy[3];
    rpy[0] = clipd(z[3] * c->angle_var[0], 2 * c->angle_var[0]);
    rpy[1] = clipd(z[4] * c->angle_var[1], 2 * c->angle_var[1]);
    rpy[2] = PI - 2 * PI * u[1];
    orc_rpy2quat(rpy, qpos + 3);
    for (int k = 0; k < 3; k++) qvel[k] = clipd(z[5 + k] * c->vel_var[k], 2 * c->vel_var[k]);
    for (int k = 0; k < 3; k++) qvel[3 + k] = clipd(z[8 + k] * c->ang_vel_var[k], 2 * c->ang_vel_var[k]);
    if (c->load) {
      for (int k = 0; k < 2; k++) qpos[7 + k] = clipd(z[11 + k] * c->pend_rp_var[k], 2 * c->pend_rp_var[k]);
      for (int k = 0; k < 2; k++) qvel[6 + k] = clipd(z[13 + k] * c->pend_vel_var[k], 2 * c->pend_vel_var[k]);
    }
  } else {
    for (int k = 0; k < 3; k++) qpos[k] = c->start_pos[k];
    double rpy[3] = {0, 0, c->start_pos[3]};
    orc_rpy2quat(rpy, qpos + 3);
    for (int k = 0; k < 6; k++) qvel[k] = 0;
    if (c->load) { qpos[7] = qpos[8] = 0; qvel[6] = qvel[7] = 0; }
  }
}

void orc_gen_params_philox(uint64_t seed, uint32_t env, uint32_t regen, const double center[6],
                           const double width[6], double difficulty, int random_params, int load,
                           double raw[6]) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, w[12];
  for (uint32_t b = 0; b < 3; b++) {
    uint32_t ctr[4] = {env, regen, b, 2u};
    orc_philox4x32(ctr, key, w + 4 * b);
  }
  for (int k = 0; k < 6; k++) {
    if (random_params) {
      uint64_t x = ((uint64_t)w[2 * k] << 32) | w[2 * k + 1];
      double uu = (double)(x >> 11) * (1.0 / 9007199254740992.0);
      double un = -width[k] + (width[k] - (-width[k])) * uu; /* numpy uniform(low, high) */
      raw[k] = center[k] + un * difficulty;
    } else {
      raw[k] = center[k];
    }
  }
  if (!load) { raw[4] = 0.0; raw[5] = 0.0; } /* BaseDroneEnv.py:212-213: pendulum * value */
}

/* ---------------------------------------------------- batched CPU baseline */
void orc_batch_step(const OrcBatchCfg *c, const OrcModel *models, const double *raw, double *qpos,
                    double *qvel, double *act, double *sensor, long *num_steps, const double *actions,
                    double *obs, double *reward, unsigned char *trunc, int threads) {
  const int nq = c->load ? 9 : 7, nv = c->load ? 8 : 6;
  const int ns = c->load ? 33 : 29;
  const int D = orc_obs_dim(c->obs_kind, ns);
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
  for (int i = 0; i < c->n; i++) {
    double ctrl[4], s[33];
    for (int r = 0; r < 4; r++) ctrl[r] = c->ctrl_map ? 0.1 + 0.9 * actions[4 * i + r] : actions[4 * i + r];
    orc_step(&models[i], c->h, c->frame_skip, qpos + nq * i, qvel + nv * i, act + 4 * i, ctrl, sensor + 3 * i);
    num_steps[i] += 1;
    if (c->obs_kind == OBS_SIMPLE) {
      orc_simple_obs(qpos + nq * i, obs + 6 * i);
      double e[3] = {obs[6 * i] - c->ref[0], obs[6 * i + 1] - c->ref[1], obs[6 * i + 2] - c->ref[2]};
      double d = sqrt(sq3(e));
      trunc[i] = d > 0.5;
      reward[i] = 0.1 - d;
      continue;
    }
    orc_drone_state(c->load, qpos + nq * i, qvel + nv * i, sensor + 3 * i, act + 4 * i, c->ref, raw + 6 * i, s);
    trunc[i] = (unsigned char)orc_truncated(s, c->ref, num_steps[i], c->max_distance, c->max_steps);
    reward[i] = orc_reward(c->reward_kind, s, ns, actions + 4 * i, num_steps[i], c->ref, c->max_distance);
    orc_obs(c->obs_kind, s, ns, c->ref, obs + (size_t)D * i);
  }
}

/* ------------------------------------------------ analytic PID cascade (8f-3) */
/* ================================================================ SURVEY 8f(1): floor contact (see qd_oracle.h) */
/* body poses in the world from qpos: R of each body and the hinge anchor */
static void floor_body_frames(const OrcModel *m, const double *qpos, double Rb[3][9], double xb[3][3]) {
  double q[4] = {qpos[3], qpos[4], qpos[5], qpos[6]};
  quat_norm(q);
  orc_quat2dcm(q, Rb[0]);
  for (int k = 0; k < 3; k++) xb[0][k] = qpos[k];
  if (m->load) {
    double Rx[9], Ry[9], t[3];
    rotx(qpos[7], Rx);
    roty(qpos[8], Ry);
    m3m(Rb[0], Rx, Rb[1]);
    m3m(Rb[1], Ry, Rb[2]);
    m3v(Rb[0], m->anchor, t);
    for (int k = 0; k < 3; k++) xb[1][k] = xb[2][k] = qpos[k] + t[k];
  }
}

static void floor_push(OrcContact *out, int *cnt, const double pos[3], double dist, int body) {
  if (*cnt >= ORC_MAX_CONTACTS) return;
  memcpy(out[*cnt].pos, pos, 3 * sizeof(double));
  out[*cnt].dist = dist;
  out[*cnt].body = body;
  (*cnt)++;
}

/* plane z = 0 against a box (mjc_PlaneBox): corners in the order x sign = bit 0, y = bit 1, z = bit 2; a corner counts when it is
 * below the plane and not above the box centre; at most four */
static void floor_box(const double c[3], const double Rg[9], const double size[3], int body, OrcContact *out, int *cnt) {
  int n = 0;
  for (int i = 0; i < 8 && n < 4; i++) {
    const double v[3] = {(i & 1 ? size[0] : -size[0]), (i & 2 ? size[1] : -size[1]), (i & 4 ? size[2] : -size[2])};
    double corner[3];
    m3v(Rg, v, corner);
    const double ldist = corner[2];
    if (c[2] + ldist > 0 || ldist > 0) continue;
    const double dist = c[2] + ldist;
    const double pos[3] = {c[0] + corner[0], c[1] + corner[1], c[2] + corner[2] - 0.5 * dist};
    floor_push(out, cnt, pos, dist, body);
    n++;
  }
}

static void floor_sphere(const double c[3], double r, int body, OrcContact *out, int *cnt) {
  const double dist = c[2] - r;
  if (dist > 0) return;
  const double pos[3] = {c[0], c[1], c[2] - r - 0.5 * dist};
  floor_push(out, cnt, pos, dist, body);
}

/* plane against a cylinder (mjc_PlaneCylinder): the deepest point of the near cap's rim, the matching point of the far cap, and two
 * more rim points of the near cap at +-120 degrees */
static void floor_cylinder(const double c[3], const double Rg[9], double radius, double hh, int body, OrcContact *out, int *cnt) {
  const double nrm[3] = {0, 0, 1};
  double axis[3] = {Rg[2], Rg[5], Rg[8]};
  double prjaxis = axis[2];
  if (prjaxis > 0) { for (int k = 0; k < 3; k++) axis[k] = -axis[k]; prjaxis = -prjaxis; }
  const double dist0 = c[2];
  double vec[3] = {axis[0] * prjaxis - nrm[0], axis[1] * prjaxis - nrm[1], axis[2] * prjaxis - nrm[2]};
  double len = sqrt(v3dot(vec, vec));
  if (len < 1e-12) { vec[0] = Rg[0] * radius; vec[1] = Rg[3] * radius; vec[2] = Rg[6] * radius; }   /* disk parallel to the plane */
  else for (int k = 0; k < 3; k++) vec[k] *= radius / len;
  const double prjvec = vec[2];
  for (int k = 0; k < 3; k++) axis[k] *= hh;
  prjaxis *= hh;
  if (dist0 + prjaxis + prjvec > 0) return;
  {
    const double d = dist0 + prjaxis + prjvec;
    const double pos[3] = {c[0] + vec[0] + axis[0], c[1] + vec[1] + axis[1], c[2] + vec[2] + axis[2] - 0.5 * d};
    floor_push(out, cnt, pos, d, body);
  }
  if (dist0 - prjaxis + prjvec <= 0) {
    const double d = dist0 - prjaxis + prjvec;
    const double pos[3] = {c[0] + vec[0] - axis[0], c[1] + vec[1] - axis[1], c[2] + vec[2] - axis[2] - 0.5 * d};
    floor_push(out, cnt, pos, d, body);
  }
  const double prjvec1 = -0.5 * prjvec;
  if (dist0 + prjaxis + prjvec1 <= 0) {
    double vec1[3];
    v3cross(vec, axis, vec1);
    double l1 = sqrt(v3dot(vec1, vec1));
    if (l1 > 1e-12) for (int k = 0; k < 3; k++) vec1[k] *= radius * sqrt(3.0) * 0.5 / l1;
    const double d = dist0 + prjaxis + prjvec1;
    for (int sgn = 1; sgn >= -1; sgn -= 2) {
      const double pos[3] = {c[0] + sgn * vec1[0] + axis[0] - 0.5 * vec[0], c[1] + sgn * vec1[1] + axis[1] - 0.5 * vec[1],
                             c[2] + sgn * vec1[2] + axis[2] - 0.5 * vec[2] - 0.5 * d};
      floor_push(out, cnt, pos, d, body);
    }
  }
}

int orc_floor_contacts(const OrcModel *m, const double *qpos, OrcContact *out) {
  double Rb[3][9], xb[3][3];
  floor_body_frames(m, qpos, Rb, xb);
  int cnt = 0;
  const double hb = 0.05, al = m->raw[1];
  const double Id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  /* a geom of body b: centre and orientation in the world */
#define GEOM(b, px, py, pz, Rl)                                   \
  double lp[3] = {px, py, pz}, c[3], Rg[9];                       \
  m3v(Rb[b], lp, c);                                              \
  for (int k = 0; k < 3; k++) c[k] += xb[b][k];                   \
  m3m(Rb[b], Rl, Rg);
  { GEOM(0, 0, 0, 0, Id) const double sz[3] = {orc_round5g(hb), orc_round5g(hb), orc_round5g(hb / 3)}; floor_box(c, Rg, sz, 0, out, &cnt); }
  { GEOM(0, orc_round5g(hb + hb / 3), 0, 0, Id)
    const double sz[3] = {orc_round5g(hb / 3), orc_round5g(0.15 * hb), orc_round5g(0.15 * hb)}; floor_box(c, Rg, sz, 0, out, &cnt); }
  for (int i = 0; i < 4; i++) {
    const double theta = i * PI / 2 - PI / 4;
    const double A = sqrt(2.0) * hb + 0.5 * al, B = sqrt(2.0) * hb + al;
    double Rz[9];
    rotz(orc_round5g(theta), Rz);
    { GEOM(0, orc_round5g(A * cos(theta)), orc_round5g(A * sin(theta)), 0, Rz)
      const double sz[3] = {orc_round5g(al / 2), orc_round5g(al / 20), orc_round5g(al / 20)}; floor_box(c, Rg, sz, 0, out, &cnt); }
    { GEOM(0, orc_round5g(B * cos(theta)), orc_round5g(B * sin(theta)), orc_round5g(0.015), Id)
      floor_cylinder(c, Rg, orc_round5g(0.01), orc_round5g(0.01), 0, out, &cnt); }
    { GEOM(0, orc_round5g(B * cos(theta)), orc_round5g(B * sin(theta)), orc_round5g(0.025), Id)
      floor_cylinder(c, Rg, orc_round5g(al / 1.5), orc_round5g(0.0025), 0, out, &cnt); }
  }
  if (m->load) {
    const double pl = m->raw[4], wm = m->raw[5];
    floor_sphere(xb[1], orc_round5g(0.02), 1, out, &cnt);
    { GEOM(2, 0, 0, orc_round5g(-pl / 2), Id) floor_cylinder(c, Rg, orc_round5g(0.005), orc_round5g(pl / 2), 2, out, &cnt); }
    { GEOM(2, 0, 0, orc_round5g(-pl), Id)
      const double bs = orc_round5g(0.1 * cbrt(wm)); const double sz[3] = {bs, bs, bs}; floor_box(c, Rg, sz, 2, out, &cnt); }
  }
#undef GEOM
  return cnt;
}

/* Jacobian of a world point x rigidly attached to body b: 3 x nv (MuJoCo free-joint convention of dyn_terms) */
static void floor_point_jac(const OrcModel *m, const double *qpos, const double Rb[3][9], const double xb[3][3], int b,
                            const double x[3], double J[3][8]) {
  memset(J, 0, 3 * 8 * sizeof(double));
  for (int j = 0; j < 3; j++) J[j][j] = 1.0;
  for (int j = 0; j < 3; j++) {
    const double a[3] = {Rb[0][j], Rb[0][3 + j], Rb[0][6 + j]}, r[3] = {x[0] - qpos[0], x[1] - qpos[1], x[2] - qpos[2]};
    double t[3];
    v3cross(a, r, t);
    for (int k = 0; k < 3; k++) J[k][3 + j] = t[k];
  }
  if (m->load && b >= 1) {
    const double r[3] = {x[0] - xb[1][0], x[1] - xb[1][1], x[2] - xb[1][2]};
    const double a1[3] = {Rb[0][0], Rb[0][3], Rb[0][6]}, a2[3] = {Rb[1][1], Rb[1][4], Rb[1][7]};
    double t[3];
    v3cross(a1, r, t);
    for (int k = 0; k < 3; k++) J[k][6] = t[k];
    if (b == 2) { v3cross(a2, r, t); for (int k = 0; k < 3; k++) J[k][7] = t[k]; }
  }
}

/* body_invweight0 (mj_setConst): at qpos0, A = J M^-1 J^T with J the 6 x nv Jacobian of the body's COM; mean diagonal of the
 * translational and of the rotational block */
static void floor_invweight(OrcModel *o) {
  double qpos[9] = {0, 0, 0, 1, 0, 0, 0, 0, 0}, qvel[8] = {0}, act[4] = {0};
  Dyn d;
  dyn_terms(o, qpos, qvel, act, &d);
  double Rb[3][9], xb[3][3];
  floor_body_frames(o, qpos, Rb, xb);
  const int nb = o->load ? 3 : 1, nv = d.nv;
  for (int b = 0; b < nb; b++) {
    double com[3];
    if (b == 0) { m3v(Rb[0], o->c0, com); }
    else if (b == 1) { memcpy(com, xb[1], sizeof com); }
    else { const double dn[3] = {0, 0, -o->lc}; m3v(Rb[2], dn, com); for (int k = 0; k < 3; k++) com[k] += xb[2][k]; }
    double Jt[3][8], Jr[3][8];
    floor_point_jac(o, qpos, Rb, xb, b, com, Jt);
    memset(Jr, 0, sizeof Jr);
    for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) Jr[k][3 + j] = Rb[0][3 * k + j];
    if (o->load && b >= 1) { for (int k = 0; k < 3; k++) Jr[k][6] = Rb[0][3 * k]; if (b == 2) for (int k = 0; k < 3; k++) Jr[k][7] = Rb[1][3 * k + 1]; }
    double tr = 0, ro = 0;
    for (int k = 0; k < 3; k++) {
      double x[8];
      spd_solve(d.M, nv, Jt[k], x);
      for (int i = 0; i < nv; i++) tr += Jt[k][i] * x[i];
      spd_solve(d.M, nv, Jr[k], x);
      for (int i = 0; i < nv; i++) ro += Jr[k][i] * x[i];
    }
    o->invweight[b][0] = tr / 3;
    o->invweight[b][1] = ro / 3;
  }
}

/* solimp (0.9, 0.95, 0.001, 0.5, 2): impedance as a function of the penetration */
static double floor_impedance(double r) {
  double x = fabs(r) / 0.001;
  if (x > 1) x = 1;
  const double y = x <= 0.5 ? 2 * x * x : 1 - 2 * (1 - x) * (1 - x);
  return 0.9 + y * (0.95 - 0.9);
}

/* adds the floor's reaction to the smooth acceleration qacc (in place); returns the number of contacts, *fz the normal force */
static int floor_solve(const OrcModel *m, const Dyn *dp, const double *qpos, const double *qvel, double h, double *qacc, double *fzp) {
  static __thread double A[4 * ORC_MAX_CONTACTS][4 * ORC_MAX_CONTACTS];
  const Dyn d = *dp;
  const int nv = d.nv;
  int ncon = 0;
    OrcContact con[ORC_MAX_CONTACTS];
    ncon = orc_floor_contacts(m, qpos, con);
    double fz = 0;
    int ne = 0;
    static __thread double Jr[4 * ORC_MAX_CONTACTS][8], MiJ[4 * ORC_MAX_CONTACTS][8], bvec[4 * ORC_MAX_CONTACTS], Rr[4 * ORC_MAX_CONTACTS],
        f[4 * ORC_MAX_CONTACTS];
    if (ncon > 0) {
      double Rb[3][9], xb[3][3];
      floor_body_frames(m, qpos, Rb, xb);
      const double mu = 1.0, tc = fmax(0.02, 2 * h), dmax = 0.95;
      const double kb = 2.0 / (dmax * tc), kk = 1.0 / (dmax * dmax * tc * tc);
      for (int c = 0; c < ncon; c++) {
        if (!(con[c].dist < 0)) continue;
        double Jp[3][8];
        floor_point_jac(m, qpos, Rb, xb, con[c].body, con[c].pos, Jp);
        const double imp = floor_impedance(con[c].dist);
        const double diag = m->invweight[con[c].body][0] * (1 + mu * mu);
        const double Rpy = 2 * mu * mu * fmax(1e-15, (1 - imp) / imp * diag);
        /* contact frame: normal z, tangents y and -x (mju_makeFrame on (0,0,1)) */
        const double dir[4][3] = {{0, mu, 1}, {0, -mu, 1}, {-mu, 0, 1}, {mu, 0, 1}};
        for (int e = 0; e < 4; e++) {
          double vel = 0;
          for (int i = 0; i < nv; i++) {
            Jr[ne][i] = dir[e][0] * Jp[0][i] + dir[e][1] * Jp[1][i] + dir[e][2] * Jp[2][i];
            vel += Jr[ne][i] * qvel[i];
          }
          const double aref = -kb * vel - kk * imp * con[c].dist;
          double a0 = 0;
          for (int i = 0; i < nv; i++) a0 += Jr[ne][i] * qacc[i];
          bvec[ne] = a0 - aref;
          Rr[ne] = Rpy;
          spd_solve(d.M, nv, Jr[ne], MiJ[ne]);
          f[ne] = 0;
          ne++;
        }
      }
      for (int i = 0; i < ne; i++)
        for (int j = 0; j < ne; j++) {
          double a = 0;
          for (int k = 0; k < nv; k++) a += Jr[i][k] * MiJ[j][k];
          A[i][j] = a;
        }
      /* projected Gauss-Seidel on  min 1/2 f^T (A + R) f + f^T b,  f >= 0 */
      for (int it = 0; it < 20000; it++) {
        double change = 0;
        for (int i = 0; i < ne; i++) {
          double g = bvec[i] + Rr[i] * f[i];
          for (int j = 0; j < ne; j++) g += A[i][j] * f[j];
          double fn = f[i] - g / (A[i][i] + Rr[i]);
          if (fn < 0) fn = 0;
          change = fmax(change, fabs(fn - f[i]));
          f[i] = fn;
        }
        if (change < 1e-13) break;
      }
      for (int i = 0; i < ne; i++) {
        for (int k = 0; k < nv; k++) qacc[k] += MiJ[i][k] * f[i];
        fz += f[i];   /* every pyramid edge has a unit normal component */
      }
    }
  *fzp = fz;
  return ncon;
}

/* mj_forward with the floor: qacc incl. the contact reaction */
int orc_forward_floor(const OrcModel *m, const double *qpos, const double *qvel, const double act[4], double h, double *qacc,
                      double *contact_force_z) {
  Dyn d;
  dyn_terms(m, qpos, qvel, act, &d);
  double rhs[8], fz = 0;
  for (int i = 0; i < d.nv; i++) rhs[i] = d.Q[i] - d.bias[i];
  spd_solve(d.M, d.nv, rhs, qacc);
  const int n = floor_solve(m, &d, qpos, qvel, h, qacc, &fz);
  if (contact_force_z) *contact_force_z = fz;
  return n;
}

int orc_step_floor(const OrcModel *m, double h, int nstep, double *qpos, double *qvel, double act[4], const double ctrl[4],
                   double sensor[3], double *contact_force_z) {
  int ncon = 0;
  for (int s = 0; s < nstep; s++) {
    Dyn d;
    dyn_terms(m, qpos, qvel, act, &d);
    const int nv = d.nv;
    double rhs[8], qacc[8], qimp[8], act_dot[4], fz = 0;
    for (int i = 0; i < nv; i++) rhs[i] = d.Q[i] - d.bias[i];
    spd_solve(d.M, nv, rhs, qacc);
    ncon = floor_solve(m, &d, qpos, qvel, h, qacc, &fz);
    if (contact_force_z) *contact_force_z = fz;
    accel_sensor(m, &d, qacc, sensor);
    for (int r = 0; r < 4; r++) act_dot[r] = (clamp01(ctrl[r]) - act[r]) / fmax(m->tau, MJMINVAL);
    if (m->load) {  /* (M + h D) qimp = M qacc, constraint forces included */
      double Mh[64], rt[8];
      memcpy(Mh, d.M, sizeof Mh);
      for (int i = 0; i < nv; i++) { rt[i] = 0; for (int j = 0; j < nv; j++) rt[i] += d.M[8 * i + j] * qacc[j]; }
      if (fz == 0) memcpy(rt, rhs, sizeof rt);  /* no contact force: exactly orc_step */
      Mh[8 * 6 + 6] += h * m->damping;
      Mh[8 * 7 + 7] += h * m->damping;
      spd_solve(Mh, nv, rt, qimp);
    } else {
      memcpy(qimp, qacc, sizeof qimp);
    }
    for (int r = 0; r < 4; r++) act[r] += h * act_dot[r];
    for (int i = 0; i < nv; i++) qvel[i] += h * qimp[i];
    for (int k = 0; k < 3; k++) qpos[k] += h * qvel[k];
    {
      double q[4] = {qpos[3], qpos[4], qpos[5], qpos[6]};
      quat_norm(q);
      double wn = sqrt(qvel[3] * qvel[3] + qvel[4] * qvel[4] + qvel[5] * qvel[5]);
      double ax[3] = {1, 0, 0};
      if (wn >= MJMINVAL) { ax[0] = qvel[3] / wn; ax[1] = qvel[4] / wn; ax[2] = qvel[5] / wn; } else wn = 0;
      double ang = h * wn, sn = sin(ang / 2);
      double qr[4] = {cos(ang / 2), ax[0] * sn, ax[1] * sn, ax[2] * sn}, qn[4];
      quat_mul(q, qr, qn);
      quat_norm(qn);
      for (int k = 0; k < 4; k++) qpos[3 + k] = qn[k];
    }
    if (m->load) { qpos[7] += h * qvel[6]; qpos[8] += h * qvel[7]; }
  }
  return ncon;
}

static double clip3(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
void orc_pid_reset(OrcPid *c) {
  memset(c, 0, sizeof *c);
  c->pos_first = c->att_first = 1;
}
void orc_pid_position(OrcPid *c, const double ref[3], const double xyz[3], double out[3]) {
  /* PositionController.py:19-34 */
  static const double P[3] = {0.4, 0.4, 0.6}, I[3] = {0.0, 0.0, 0.01}, D[3] = {0.15, 0.15, 0.2};
  const double dt = 0.02;
  for (int k = 0; k < 3; k++) {
    double e = clip3(ref[k] - xyz[k], -2, 2);
    if (c->pos_first) c->pos_prev[k] = e;
    double ed = (e - c->pos_prev[k]) / dt;
    c->pos_prev[k] = e;
    c->pos_i[k] = clip3(c->pos_i[k] + dt * e, -1, 1);
    double o = P[k] * e + I[k] * c->pos_i[k] + D[k] * ed;
    out[k] = k < 2 ? clip3(o, -0.5, 0.5) : clip3(o, -2, 2);
  }
  c->pos_first = 0;
}
void orc_pid_tilts2rpy(const double pa[3], double heading, double rpyz[4]) {
  /* AttitudeController.py:24-39.  Rd = [y x z, y, z] with y = z x heading is NOT orthonormal: |y| = |y x z| = s < 1,
   * i.e. Rd = Q diag(s, s, 1) with Q orthonormal and right-handed.  scipy (1.15.3, the version the golden vectors
   * were generated with) orthogonalises from_matrix input by the orthogonal-Procrustes / polar factor, which for
   * Q * (positive diagonal) is exactly Q: normalise the y column.  (scipy < 1.4-era releases applied Markley's
   * formula to the raw matrix instead; the reference pins no scipy version, the golden vectors pin this one.) */
  double zacc = pa[2] + 9.81;
  double t[3] = {tan(pa[0]), tan(pa[1]), 1.0};
  double tn = sqrt(t[0] * t[0] + t[1] * t[1] + 1.0);
  double z[3] = {t[0] / tn, t[1] / tn, 1.0 / tn}, hv[3] = {cos(heading), sin(heading), 0.0}, y[3], x[3];
  v3cross(z, hv, y);
  double yn = sqrt(v3dot(y, y));
  for (int k = 0; k < 3; k++) y[k] /= yn;
  v3cross(y, z, x);
  double R[9] = {x[0], y[0], z[0], x[1], y[1], z[1], x[2], y[2], z[2]}, q[4];
  orc_dcm2quat(R, q);
  orc_quat2rpy(q, rpyz);
  rpyz[3] = tn * fabs(zacc);
}
void orc_pid_attitude(OrcPid *c, const double rpyz[4], const double rpy[3], double mass, double motor_force,
                      double ctrl[4]) {
  /* AttitudeController.py:41-55 */
  static const double P[3] = {2, 2, 0.1}, I[3] = {0, 0, 0}, D[3] = {0.2, 0.2, 0};
  static const double mixer[4][3] = {{-1, -1, 1}, {1, -1, -1}, {1, 1, 1}, {-1, 1, -1}};
  const double dt = 0.02;
  double a[3];
  for (int k = 0; k < 3; k++) {
    double e = rpyz[k] - rpy[k];
    if (c->att_first) c->att_prev[k] = e;
    double ed = (e - c->att_prev[k]) / dt;
    c->att_prev[k] = e;
    c->att_i[k] = clip3(c->att_i[k] + dt * e, -1, 1);
    a[k] = P[k] * e + I[k] * c->att_i[k] + D[k] * ed;
  }
  c->att_first = 0;
  for (int m = 0; m < 4; m++) {
    double f = mixer[m][0] * a[0] + mixer[m][1] * a[1] + mixer[m][2] * a[2] + 0.25 * rpyz[3] * mass;
    ctrl[m] = clip3(f / motor_force, 0, 1);
  }
}
void orc_pid_action(OrcPid *c, const double ref[4], const double xyz[3], const double rpy[3], double mass,
                    double motor_force, double action[4]) {
  double pa[3], rpyz[4];
  orc_pid_position(c, ref, xyz, pa);
  orc_pid_tilts2rpy(pa, ref[3], rpyz);
  orc_pid_attitude(c, rpyz, rpy, mass, motor_force, action);
  for (int m = 0; m < 4; m++) action[m] = clip3(action[m] - 0.1, 0, 1);
}

/* ------------------------------------------------ waypoint generators (8f-4) */
void orc_trajectory_point(int mode, const double p[4], const double start[4], const double end[4], double dt, long k,
                          double out[4]) {
  const double t = (double)k * dt; /* numpy arange: start + k * step */
  if (mode == 1) {                 /* evaluation.py:135-138 */
    out[0] = p[1] * cos(2 * PI * p[0] * t); out[1] = p[1] * sin(2 * PI * p[0] * t); out[2] = p[2]; out[3] = 0;
  } else if (mode == 2) {          /* evaluation.py:141-144 */
    for (int c = 0; c < 4; c++) out[c] = t < p[0] ? start[c] : end[c];
  } else {                         /* evaluation.py:147-152 */
    for (int c = 0; c < 4; c++)
      out[c] = t < p[0] ? start[c] : start[c] + (t - p[0]) / (p[1] - p[0]) * (end[c] - start[c]);
  }
}
