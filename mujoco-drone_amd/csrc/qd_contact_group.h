// qd_contact_group.h -- the floor-contact solve of qd_contact.h as a LANE-GROUP computation (device only; the
// north_star's "wavefront shuffle reductions for the constraint / contact solve").
//
// qd_contact.h solves the contact problem of one env inside one lane: a contact list of up to 40 points in scratch, a Newton
// method whose every cost evaluation re-derives every contact row, 6 x 6 / 8 x 8 factorisations in float64 -- 1.8 KB of scratch
// and 155 us per step of 4096 load-model envs when all of them touch the floor.  Same convex problem here, same constants,
// same Newton method, other mapping:
//   * the lanes of a wavefront whose env can reach the floor (height test) are compacted; 8 envs at a time get a GROUP OF 8
//     LANES each (8 groups = one wavefront);
//   * contact generation: the env's 14 (single body) / 17 (with link, tether rod and load box) geoms are dealt to the 8 lanes,
//     each lane tests its two or three geoms against the plane; the contact points go to a list in LDS at offsets from an
//     8-lane prefix sum (two passes over the same code: count, then store -- no per-lane staging, a deterministic order);
//   * the contacts are dealt back to the lanes round-robin; in a Newton iteration each lane forms the pyramid-edge rows of ITS
//     contacts (4 edges per contact point) and accumulates its share of the gradient and of J^T D J; the 8 partial sums are added
//     by a 3-stage xor butterfly of cross-lane shuffles (every lane ends with the same bits: a + b and b + a are the same float);
//   * the mass matrix, the Cholesky factorisation, the Newton step and the line-search bookkeeping are replicated in the 8
//     lanes (identical inputs, identical results), the line search's cost is again a partial sum + butterfly;
//   * the result goes back to the env's own lane through LDS.
// No scratch, no call; contact rows are never stored (a row is a function of the contact point: two cross products).
//
// One formulation for both models: MuJoCo's generalised coordinates, nv = 6 (free joint) or 8 (+ two hinges); the single-body
// solve of qd_contact.h (COM-centred coordinates, diagonal mass matrix) is the same convex problem in other coordinates, so the
// minimiser is the same; tests/test_gpu_floor.py holds both to the float64 oracle's own solver (dual projected Gauss-Seidel).
// PARITY UNPINNED exactly as qd_contact.h says: contact rules and constants restate MuJoCo's from memory.
#pragma once
#include "qd_contact.h"

namespace qd {

#ifdef QD_STAMPS
// diagnostic build: cycle stamps of workgroup 0 (k_step_floor: 0..7 owner wave; cg_solve: 8..15 per wave w at 8 + 2 w ...)
__device__ unsigned long long qd_sfstamps[64];
#define SF_STAMP(k)                                                                 \
  do {                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                              \
    unsigned long long t_;                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
    __builtin_amdgcn_sched_barrier(0);                                              \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) qd_sfstamps[(k)] = t_;          \
  } while (0)
#else
#define SF_STAMP(k)
#endif

constexpr int CG_ENVS = 8;   // envs per pass of a wavefront = groups of 8 lanes

struct CgRecord {             // one touching env, written by its own lane (float64 throughout, like qd_contact.h)
  double p[3], R[9], R1[9], R2[9], xa[3];   // TreePose
  double qv[8], a0[8];                      // generalised velocity, unconstrained acceleration (explicit damping)
  double m0, c0z, I0x, I0y, I0z, m2, lc, I2t, I2a;
  // geom placement and sizes as they reach MuJoCo (%.5g), derived once by the env's own lane
  double pa, pm, arm_half, arm_thin, prop_r, rod_z, box_z, rod_half, box_half;
  double tran[3];                           // translational body_invweight0 of core, link, tether + load (of the parameter set)
};
constexpr int CG_BLOCK_ENVS = 32;   // envs per workgroup of k_step_floor (four wavefronts: four passes of 8 envs run side by side)
constexpr int CG_LANE_CONTACTS = 12;   // a lane tests at most three geoms, each yields at most four points
struct CgWave {               // the work area of one wavefront's pass: 8 envs
  double con[CG_ENVS][8][CG_LANE_CONTACTS][4];   // per lane of the group: contact point (world), signed distance
  int body[CG_ENVS][8][CG_LANE_CONTACTS];        // 0 core, 1 link, 2 tether + load
  double mm[CG_ENVS][36];                // the env's mass matrix (packed lower triangle): read back where it is needed instead of
                                         // living in 72 registers through the Newton iterations
};
struct CgLds {                // per workgroup
  CgRecord rec[CG_BLOCK_ENVS];           // the touching envs, compacted (rank order)
  double res[CG_BLOCK_ENVS][17];         // explicit accelerations (8), damping-implicit accelerations (8), normal force
  CgWave w[4];
};

// The sum over the 8 lanes of a group, the same bits in every lane: three data-parallel-primitive stages -- swap neighbours,
// swap pairs (quad_perm), mirror the half row (lane i <-> 7 - i: the other quad's sum) -- each a + b = b + a on both sides.
// (Cross-lane operands without an LDS round trip: ds_bpermute-based __shfl_xor cost 2 LDS instructions + a wait per stage.)
template <int CTRL>
__device__ __forceinline__ double cg_dpp(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double cg_sum8(double v) {
  v += cg_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += cg_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += cg_dpp<0x141>(v);   // row_half_mirror
  return v;
}

struct CgCount {   // pass 1: how many points, how many of them penetrating
  int n, nact, cur;
  __device__ __forceinline__ void push(double, double, double, double dist) { n++; nact += dist < 0.0 ? 1 : 0; }
};
struct CgStore {   // into the lane's own list (the lanes of a group keep the contacts of the geoms they tested)
  double (*dst)[4];
  int* body;
  int at, nact, cur;
  __device__ __forceinline__ void push(double px, double py, double pz, double dist) {
    if (at < CG_LANE_CONTACTS) { dst[at][0] = px; dst[at][1] = py; dst[at][2] = pz; dst[at][3] = dist; body[at] = cur; at++; nact += dist < 0.0 ? 1 : 0; }
  }
};

// geom `gi` of the env against the plane z = 0: 0 core box, 1 front marker, 2..5 arm boxes, 6..9 motors, 10..13 propeller disks
// (env_gen.py:41-61, order of qd_contact.h's contact_generate regrouped), 14 link sphere, 15 tether rod, 16 load box (:66-72)
template <class CS>
__device__ __forceinline__ void cg_geom(CS& cs, int gi, const CgRecord& r) {
  const double hb = 0.05;
  const double Id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  auto place = [&](double lx, double ly, double lz, const double Rl[9], double c[3], double Rg[9]) {
    c[0] = r.p[0] + r.R[0] * lx + r.R[1] * ly + r.R[2] * lz;
    c[1] = r.p[1] + r.R[3] * lx + r.R[4] * ly + r.R[5] * lz;
    c[2] = r.p[2] + r.R[6] * lx + r.R[7] * ly + r.R[8] * lz;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) Rg[3 * i + j] = r.R[3 * i] * Rl[j] + r.R[3 * i + 1] * Rl[3 + j] + r.R[3 * i + 2] * Rl[6 + j];
  };
  double c[3], Rg[9];
  cs.cur = 0;
  if (gi == 0) {
    place(0, 0, 0, Id, c, Rg);
    contact_box(cs, c, Rg, 0.05, 0.05, 0.016667);                    // round5(hb), round5(hb / 3)
  } else if (gi == 1) {
    place(0.066667, 0, 0, Id, c, Rg);                                 // round5(hb + hb / 3)
    contact_box(cs, c, Rg, 0.016667, 0.0075, 0.0075);                 // round5(hb / 3), round5(0.15 hb)
  } else if (gi < 14) {
    const int i = (gi - 2) & 3;
    const double sgx = i < 2 ? 1.0 : -1.0, sgy = (i == 1 || i == 2) ? 1.0 : -1.0;
    if (gi < 6) {
      // cos / sin of theta_i = i pi/2 - pi/4 as printed with 5 digits: -0.7854, 0.7854, 2.3562, 3.927
      const double ct = i < 2 ? 0.70710548251123628 : (i == 2 ? -0.70711067719817011 : -0.70710028778613998);
      const double st = i == 0 ? -0.70710807985947355 : (i == 1 ? 0.70710807985947355 : (i == 2 ? 0.70710288515345854 : -0.70711327452732631));
      const double Rz[9] = {ct, -st, 0, st, ct, 0, 0, 0, 1};
      place(sgx * r.pa, sgy * r.pa, 0, Rz, c, Rg);
      contact_box(cs, c, Rg, r.arm_half, r.arm_thin, r.arm_thin);
    } else if (gi < 10) {
      place(sgx * r.pm, sgy * r.pm, 0.015, Id, c, Rg);
      contact_cylinder(cs, c, Rg, 0.01, 0.01);
    } else {
      place(sgx * r.pm, sgy * r.pm, 0.025, Id, c, Rg);
      contact_cylinder(cs, c, Rg, r.prop_r, 0.0025);
    }
  } else if (gi == 14) {
    cs.cur = 1;
    const double dist = r.xa[2] - Const::r1;
    if (dist <= 0.0) cs.push(r.xa[0], r.xa[1], r.xa[2] - Const::r1 - 0.5 * dist, dist);
  } else {
    cs.cur = 2;
    const double z = gi == 15 ? r.rod_z : r.box_z;
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = r.xa[k] + r.R2[3 * k + 2] * z;
    if (gi == 15) contact_cylinder(cs, c, r.R2, 0.005, r.rod_half);
    else contact_box(cs, c, r.R2, r.box_half, r.box_half, r.box_half);
  }
  (void)hb;
}

// round5() of qd_model.h with the decimal exponent found by comparisons instead of floor(log10()) (a float64 log10 is ~150
// instructions, and an env's nine rounded sizes are derived in every substep it touches the floor); same arithmetic after that,
// for the magnitudes geometry has (1e-7 .. 1e7)
__device__ __forceinline__ double cg_round5(double x) {
  if (x == 0.0) return 0.0;
  const double ax = fabs(x);
  // k = 4 - floor(log10(ax)), found on exact powers of ten (literals: 10^-j is the nearest double, the one log10 maps to -j)
  int k = 4;
  if (ax >= 1e1) k = 3;
  if (ax >= 1e2) k = 2;
  if (ax >= 1e3) k = 1;
  if (ax >= 1e4) k = 0;
  if (ax >= 1e5) k = -1;
  if (ax >= 1e6) k = -2;
  if (ax < 1e0) k = 5;
  if (ax < 1e-1) k = 6;
  if (ax < 1e-2) k = 7;
  if (ax < 1e-3) k = 8;
  if (ax < 1e-4) k = 9;
  if (ax < 1e-5) k = 10;
  if (ax < 1e-6) k = 11;
  // 10^|k| exactly, by selects (no table in memory)
  const int m = k < 0 ? -k : k;
  double p = 1.0;
  if (m & 1) p *= 1e1;
  if (m & 2) p *= 1e2;
  if (m & 4) p *= 1e4;
  if (m & 8) p *= 1e8;
  const double r = k >= 0 ? rint(ax * p) / p : rint(ax / p) * p;
  return x < 0 ? -r : r;
}

constexpr int cg_tri(int i, int j) { return i * (i + 1) / 2 + j; }   // packed lower triangle, j <= i

// mass matrix of the tree in MuJoCo's coordinates, packed lower triangle (qd_contact.h: tree_mass_matrix, unrolled for registers)
template <int NV>
__device__ __forceinline__ void cg_mass_matrix(const CgRecord& r, const double p[3], const double R[9], const double R1[9], const double R2[9],
                                               const double xa[3], double Mm[NV * (NV + 1) / 2]) {
#pragma unroll
  for (int k = 0; k < NV * (NV + 1) / 2; k++) Mm[k] = 0.0;
  constexpr int NB = NV == 8 ? 3 : 1;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const double mass = b == 0 ? r.m0 : (b == 1 ? Const::m1 : r.m2);
    const double Ix = b == 0 ? r.I0x : (b == 1 ? Const::I1 : r.I2t), Iy = b == 0 ? r.I0y : (b == 1 ? Const::I1 : r.I2t),
                 Iz = b == 0 ? r.I0z : (b == 1 ? Const::I1 : r.I2a);
    const double* Rb = b == 0 ? R : (b == 1 ? R1 : R2);
    double c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = b == 0 ? p[k] + R[3 * k + 2] * r.c0z : (b == 1 ? xa[k] : xa[k] - R2[3 * k + 2] * r.lc);
    // COM Jacobian (3 x NV) and rotational Jacobian in the body's own axes
    double Jp[3][NV], Jw[3][NV];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int j = 0; j < NV; j++) { Jp[k][j] = (j == k) ? 1.0 : 0.0; Jw[k][j] = 0.0; }
    const double rr[3] = {c[0] - p[0], c[1] - p[1], c[2] - p[2]};
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double ax = R[j], ay = R[3 + j], az = R[6 + j];
      Jp[0][3 + j] = ay * rr[2] - az * rr[1]; Jp[1][3 + j] = az * rr[0] - ax * rr[2]; Jp[2][3 + j] = ax * rr[1] - ay * rr[0];
#pragma unroll
      for (int k = 0; k < 3; k++) Jw[k][3 + j] = Rb[k] * ax + Rb[3 + k] * ay + Rb[6 + k] * az;
    }
    if constexpr (NV == 8) if (b >= 1) {
      const double ra[3] = {c[0] - xa[0], c[1] - xa[1], c[2] - xa[2]};
      {
        const double ax = R[0], ay = R[3], az = R[6];
        Jp[0][6] = ay * ra[2] - az * ra[1]; Jp[1][6] = az * ra[0] - ax * ra[2]; Jp[2][6] = ax * ra[1] - ay * ra[0];
#pragma unroll
        for (int k = 0; k < 3; k++) Jw[k][6] = Rb[k] * ax + Rb[3 + k] * ay + Rb[6 + k] * az;
      }
      if (b == 2) {
        const double ax = R1[1], ay = R1[4], az = R1[7];
        Jp[0][7] = ay * ra[2] - az * ra[1]; Jp[1][7] = az * ra[0] - ax * ra[2]; Jp[2][7] = ax * ra[1] - ay * ra[0];
#pragma unroll
        for (int k = 0; k < 3; k++) Jw[k][7] = Rb[k] * ax + Rb[3 + k] * ay + Rb[6 + k] * az;
      }
    }
#pragma unroll
    for (int i = 0; i < NV; i++)
#pragma unroll
      for (int j = 0; j <= i; j++)
        Mm[cg_tri(i, j)] += mass * (Jp[0][i] * Jp[0][j] + Jp[1][i] * Jp[1][j] + Jp[2][i] * Jp[2][j]) +
                            Ix * Jw[0][i] * Jw[0][j] + Iy * Jw[1][i] * Jw[1][j] + Iz * Jw[2][i] * Jw[2][j];
  }
}

// in-place Cholesky of a packed lower triangle and the solve with it (every index a compile-time constant: registers).
// The diagonal holds 1 / L_ii: one square root and one division per column instead of a division per entry (float64 divisions
// are ~25 instructions each; 44 of them per Newton iteration were a third of the iteration).
template <int NV>
__device__ __forceinline__ void cg_chol(double A[NV * (NV + 1) / 2]) {
#pragma unroll
  for (int j = 0; j < NV; j++) {
    double s = A[cg_tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; k++) s -= A[cg_tri(j, k)] * A[cg_tri(j, k)];
    const double inv = 1.0 / sqrt(s);
    A[cg_tri(j, j)] = inv;
#pragma unroll
    for (int i = j + 1; i < NV; i++) {
      double t = A[cg_tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; k++) t -= A[cg_tri(i, k)] * A[cg_tri(j, k)];
      A[cg_tri(i, j)] = t * inv;
    }
  }
}
template <int NV>
__device__ __forceinline__ void cg_chol_solve(const double Lc[NV * (NV + 1) / 2], const double b[NV], double x[NV]) {
  double y[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= Lc[cg_tri(i, k)] * y[k];
    y[i] = s * Lc[cg_tri(i, i)];
  }
#pragma unroll
  for (int i = NV - 1; i >= 0; i--) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < NV; k++) s -= Lc[cg_tri(k, i)] * x[k];
    x[i] = s * Lc[cg_tri(i, i)];
  }
}

// The contact solve in three steps (k_step_floor, qd_step_floor.hip):
//   cg_publish   the lane that owns a touching env describes it in G.rec[rank] (rank = its place among the workgroup's touching envs)
//   cg_solve     a whole wavefront solves 8 of them, G.rec[base .. base + 7], one per group of 8 lanes -> G.res[base + group]
//   cg_collect   the owning lane takes the constrained accelerations (if the floor pushes at all)
// fc: the parameter set's floor constants (FLOOR_CONSTS planes of the arena, k_floor_consts)
template <bool LOAD>
__device__ __forceinline__ void cg_publish(CgRecord& r, const Model<float>& M, const State<float>& s, const double* fc, const Accel<float>& ex) {
  TreePose P;
  const double p[3] = {(double)s.px, (double)s.py, (double)s.pz}, q[4] = {(double)s.qw, (double)s.qx, (double)s.qy, (double)s.qz};
  tree_pose(p, q, LOAD ? (double)s.th1 : 0.0, LOAD ? (double)s.th2 : 0.0, P);
#pragma unroll
  for (int k = 0; k < 3; k++) { r.p[k] = P.p[k]; r.xa[k] = P.xa[k]; }
#pragma unroll
  for (int k = 0; k < 9; k++) { r.R[k] = P.R[k]; r.R1[k] = P.R1[k]; r.R2[k] = P.R2[k]; }
  r.qv[0] = s.vx; r.qv[1] = s.vy; r.qv[2] = s.vz; r.qv[3] = s.wx; r.qv[4] = s.wy; r.qv[5] = s.wz; r.qv[6] = s.thd1; r.qv[7] = s.thd2;
  r.a0[0] = ex.lin.x; r.a0[1] = ex.lin.y; r.a0[2] = ex.lin.z; r.a0[3] = ex.ang.x; r.a0[4] = ex.ang.y; r.a0[5] = ex.ang.z;
  r.a0[6] = ex.thdd1; r.a0[7] = ex.thdd2;
  r.m0 = M.m0; r.c0z = M.c0z; r.I0x = M.I0x; r.I0y = M.I0y; r.I0z = M.I0z; r.m2 = M.m2; r.lc = M.lc; r.I2t = M.I2t; r.I2a = M.I2a;
  r.pa = fc[FC_PA]; r.pm = fc[FC_PM]; r.arm_half = fc[FC_ARM_HALF]; r.arm_thin = fc[FC_ARM_THIN]; r.prop_r = fc[FC_PROP_R];
  r.rod_z = fc[FC_ROD_Z]; r.box_z = fc[FC_BOX_Z]; r.rod_half = fc[FC_ROD_HALF]; r.box_half = fc[FC_BOX_HALF];
  r.tran[0] = fc[FC_TRAN0]; r.tran[1] = fc[FC_TRAN1]; r.tran[2] = fc[FC_TRAN2];
}

// What a parameter set fixes of the floor-contact problem, derived ONCE per parameter set (k_floor_consts) instead of in every
// substep an env touches the floor: the geom sizes and placements as they reach MuJoCo (%.5g), the reach below the origin, and
// the translational body_invweight0 of the three bodies -- J M^-1 J^T of each body's COM at qpos0 (identity attitude, tether straight
// down): one 8 x 8 factorisation and nine solves (4.5 k of a substep's 98 k cycles; the nine roundings another ~5 k).
template <bool LOAD>
__device__ __forceinline__ void cg_floor_consts(const Model<float>& M, double arm_len, double pend_len, double weight_mass, double fc[13]) {
  constexpr int NV = LOAD ? 8 : 6, NT = NV * (NV + 1) / 2;
  const double sq2 = 1.4142135623730951, cs45 = 0.70710678118654752440;
  fc[FC_PA] = cg_round5((sq2 * 0.05 + 0.5 * arm_len) * cs45);
  fc[FC_PM] = cg_round5((sq2 * 0.05 + arm_len) * cs45);
  fc[FC_ARM_HALF] = cg_round5(arm_len / 2); fc[FC_ARM_THIN] = cg_round5(arm_len / 20); fc[FC_PROP_R] = cg_round5(arm_len / 1.5);
  fc[FC_ROD_Z] = LOAD ? cg_round5(-pend_len / 2) : 0.0; fc[FC_BOX_Z] = LOAD ? cg_round5(-pend_len) : 0.0;
  fc[FC_ROD_HALF] = LOAD ? cg_round5(pend_len / 2) : 0.0; fc[FC_BOX_HALF] = LOAD ? cg_round5(0.1 * cbrt(weight_mass)) : 0.0;
  // nothing of the drone reaches further than this below its origin (qd_contact.h: floor_contact / floor_contact_tree)
  double reach = 1.4142135623730951 * 0.05 + arm_len * (1.0 + 1.0 / 1.5) + 0.03;
  if (LOAD) reach += pend_len + 1.7320508075688772 * fc[FC_BOX_HALF];
  fc[FC_REACH] = reach;
  fc[FC_TRAN0] = 1.0 / (double)M.m0; fc[FC_TRAN1] = 0.0; fc[FC_TRAN2] = 0.0;
  if constexpr (LOAD) {
    CgRecord r;
    r.m0 = M.m0; r.c0z = M.c0z; r.I0x = M.I0x; r.I0y = M.I0y; r.I0z = M.I0z; r.m2 = M.m2; r.lc = M.lc; r.I2t = M.I2t; r.I2a = M.I2a;
    const double z3[3] = {0, 0, 0}, I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, xa0[3] = {0, 0, Const::anchor_z};
    double L0[NT];
    cg_mass_matrix<NV>(r, z3, I9, I9, I9, xa0, L0);
    cg_chol<NV>(L0);
    double part[3] = {0.0, 0.0, 0.0};
#pragma unroll 1
    for (int q9 = 0; q9 < 9; q9++) {
      const int b = q9 / 3, k = q9 - 3 * b;
      const double z = b == 0 ? r.c0z : (b == 1 ? Const::anchor_z : Const::anchor_z - r.lc), za = z - Const::anchor_z;
      double Jr[NV], xs[NV];
#pragma unroll
      for (int j = 0; j < NV; j++) Jr[j] = (j == k) ? 1.0 : 0.0;
      // (e_j x r)_k with r = (0, 0, z): e_x x r = (0, -z, 0), e_y x r = (z, 0, 0), e_z x r = 0
      if (k == 1) Jr[3] = -z;
      if (k == 0) Jr[4] = z;
      if (b >= 1 && k == 1) Jr[6] = -za;   // hinge x, from the anchor
      if (b == 2 && k == 0) Jr[7] = za;    // hinge y
      cg_chol_solve<NV>(L0, Jr, xs);
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < NV; j++) t += Jr[j] * xs[j];
      part[b] += t;
    }
    fc[FC_TRAN0] = part[0] / 3.0; fc[FC_TRAN1] = part[1] / 3.0; fc[FC_TRAN2] = part[2] / 3.0;
  }
}

template <bool LOAD>
__device__ __forceinline__ void cg_solve(CgLds& G, CgWave& W, int base, int total, double h) {
  constexpr int NV = LOAD ? 8 : 6, NT = NV * (NV + 1) / 2, NG = LOAD ? 17 : 14;
  const int lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7;
  if (base + grp < total) {
    const CgRecord& r = G.rec[base + grp];
    // contact generation: ONE pass, every lane into its own list.  The geoms are dealt so that the lanes of a wavefront run the
    // same code together: first the boxes (core, marker, four arms, load box; the link sphere rides along on the eighth lane),
    // then the eight rotor cylinders, then the tether rod on one lane -- a box test and two cylinder tests per lane instead of
    // three rounds of both (lanes in different branches execute one after the other).
    CgStore st;
    st.dst = W.con[grp][sub]; st.body = W.body[grp][sub]; st.at = 0; st.nact = 0; st.cur = 0;
    {
      const int g0 = sub < 6 ? sub : (sub == 6 ? 16 : 14);            // boxes 0..5, the load box, the link sphere
      if (g0 < NG) cg_geom(st, g0, r);
      cg_geom(st, 6 + sub, r);                                          // motors 6..9, propeller disks 10..13
      if (NG > 15 && sub == 0) cg_geom(st, 15, r);                      // the tether rod
    }
    const int myn = st.at;
    if (threadIdx.x < 64) SF_STAMP(8);
    int nact = st.nact;
    nact += __shfl_xor(nact, 1); nact += __shfl_xor(nact, 2); nact += __shfl_xor(nact, 4);
    int ncon = myn;   // (statistics only)
    ncon += __shfl_xor(ncon, 1); ncon += __shfl_xor(ncon, 2); ncon += __shfl_xor(ncon, 4);
    double fz = 0.0;
    double qa[NV], qi[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) qa[k] = qi[k] = 0.0;
    if (nact > 0) {   // (uniform in the group)
      // replicated per group: pose, mass matrix, body_invweight0 (translational) of the three bodies
      double p[3], R[9], xa[3], a2[3], ab[NV];
#pragma unroll
      for (int k = 0; k < 3; k++) { p[k] = r.p[k]; xa[k] = r.xa[k]; a2[k] = r.R1[3 * k + 1]; }   // a2: hinge-y axis (world)
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = r.R[k];
      {
        double R1[9], R2[9], Mm[NT];
#pragma unroll
        for (int k = 0; k < 9; k++) { R1[k] = r.R1[k]; R2[k] = r.R2[k]; }
        cg_mass_matrix<NV>(r, p, R, R1, R2, xa, Mm);
        if (sub == 0) {
#pragma unroll
          for (int k = 0; k < NT; k++) W.mm[grp][k] = Mm[k];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const double* Mm = W.mm[grp];
      if (threadIdx.x < 64) SF_STAMP(9);
      const double tran[3] = {r.tran[0], r.tran[1], r.tran[2]};   // body_invweight0: a function of the parameter set (cg_floor_consts)
      if (threadIdx.x < 64) SF_STAMP(10);
      const double mu = 1.0, tc = h * 2.0 > 0.02 ? h * 2.0 : 0.02, dmax = 0.95;
      const double kb = 2.0 / (dmax * tc), kk = 1.0 / (dmax * dmax * tc * tc);
#pragma unroll
      for (int k = 0; k < NV; k++) ab[k] = r.a0[k] + kb * r.qv[k];   // a row's residual only ever sees a0 + kb qv
      // my contacts: sub, sub + 8, ...  A contact's rows at acceleration change x: the point's acceleration / velocity
      //   y = J_p (a0 + x), v = J_p qv;  edge e: val = u_e . (y + kb v) + kk imp dist,  u = (0, +-mu, 1), (-+mu, 0, 1)
      auto point_jac = [&](const double xp[3], int b, double Jp[3][NV]) {
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
          for (int j = 0; j < NV; j++) Jp[k][j] = (j == k) ? 1.0 : 0.0;
        const double rr[3] = {xp[0] - p[0], xp[1] - p[1], xp[2] - p[2]};
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const double ax = R[j], ay = R[3 + j], az = R[6 + j];
          Jp[0][3 + j] = ay * rr[2] - az * rr[1]; Jp[1][3 + j] = az * rr[0] - ax * rr[2]; Jp[2][3 + j] = ax * rr[1] - ay * rr[0];
        }
        if constexpr (NV == 8) {
          const double ra[3] = {xp[0] - xa[0], xp[1] - xa[1], xp[2] - xa[2]};
          const double k1 = b >= 1 ? 1.0 : 0.0, k2 = b == 2 ? 1.0 : 0.0;
          {
            const double ax = R[0], ay = R[3], az = R[6];
            Jp[0][6] = k1 * (ay * ra[2] - az * ra[1]); Jp[1][6] = k1 * (az * ra[0] - ax * ra[2]); Jp[2][6] = k1 * (ax * ra[1] - ay * ra[0]);
          }
          {
            const double ax = a2[0], ay = a2[1], az = a2[2];
            Jp[0][7] = k2 * (ay * ra[2] - az * ra[1]); Jp[1][7] = k2 * (az * ra[0] - ax * ra[2]); Jp[2][7] = k2 * (ax * ra[1] - ay * ra[0]);
          }
        }
      };
      // the part of the cost that is mine at acceleration change xx (0.5 x^T M x is added once, replicated)
      unsigned long long cm = 0ull;   // my active edges and my share of the normal force at the last point my_cost evaluated
      double cfz = 0.0;
      auto my_cost = [&](const double xx[NV]) {
        double cst = 0.0;
        cm = 0ull; cfz = 0.0;
        for (int c = 0; c < myn; c++) {
          const double dist = W.con[grp][sub][c][3];
          if (!(dist < 0.0)) continue;
          const double xp[3] = {W.con[grp][sub][c][0], W.con[grp][sub][c][1], W.con[grp][sub][c][2]};
          const int b = W.body[grp][sub][c];
          double Jp[3][NV];
          point_jac(xp, b, Jp);
          double y[3];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < NV; j++) acc += Jp[k][j] * (ab[j] + xx[j]);
            y[k] = acc;
          }
          const double imp = contact_impedance(dist);
          const double cn = y[2] + kk * imp * dist;
          double Rr = 2.0 * mu * mu * (1.0 - imp) / imp * (1.0 + mu * mu) * (b == 0 ? tran[0] : (b == 1 ? tran[1] : tran[2]));
          if (Rr < 1e-15) Rr = 1e-15;
          const double D = 1.0 / Rr;
          const double v0 = cn + mu * y[1], v1 = cn - mu * y[1], v2 = cn - mu * y[0], v3 = cn + mu * y[0];
          if (v0 < 0.0) { cst += 0.5 * D * v0 * v0; cm |= 1ull << (4 * c + 0); cfz -= D * v0; }
          if (v1 < 0.0) { cst += 0.5 * D * v1 * v1; cm |= 1ull << (4 * c + 1); cfz -= D * v1; }
          if (v2 < 0.0) { cst += 0.5 * D * v2 * v2; cm |= 1ull << (4 * c + 2); cfz -= D * v2; }
          if (v3 < 0.0) { cst += 0.5 * D * v3 * v3; cm |= 1ull << (4 * c + 3); cfz -= D * v3; }
        }
        return cst;
      };
      auto quad = [&](const double xx[NV]) {   // 0.5 x^T M x
        double q2 = 0.0;
#pragma unroll
        for (int i = 0; i < NV; i++) {
          double mx = 0.0;
#pragma unroll
          for (int j = 0; j < NV; j++) mx += Mm[i >= j ? cg_tri(i, j) : cg_tri(j, i)] * xx[j];
          q2 += 0.5 * xx[i] * mx;
        }
        return q2;
      };
      double xk[NV];
#pragma unroll
      for (int k = 0; k < NV; k++) xk[k] = 0.0;
      double ck = cg_sum8(my_cost(xk));
      if (threadIdx.x < 64) SF_STAMP(11);
      for (int it = 0; it < 80; it++) {
        double g[NV], H[NT];
#pragma unroll
        for (int k = 0; k < NV; k++) g[k] = 0.0;
#pragma unroll
        for (int k = 0; k < NT; k++) H[k] = 0.0;
        double fzp = 0.0;
        unsigned long long am = 0ull;   // my active edges at xk (bit 4 c + e)
        for (int c = 0; c < myn; c++) {
          const double dist = W.con[grp][sub][c][3];
          if (!(dist < 0.0)) continue;
          const double xp[3] = {W.con[grp][sub][c][0], W.con[grp][sub][c][1], W.con[grp][sub][c][2]};
          const int b = W.body[grp][sub][c];
          double Jp[3][NV];
          point_jac(xp, b, Jp);
          double y[3];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < NV; j++) acc += Jp[k][j] * (ab[j] + xk[j]);
            y[k] = acc;
          }
          const double imp = contact_impedance(dist);
          const double cn = y[2] + kk * imp * dist;
          double Rr = 2.0 * mu * mu * (1.0 - imp) / imp * (1.0 + mu * mu) * (b == 0 ? tran[0] : (b == 1 ? tran[1] : tran[2]));
          if (Rr < 1e-15) Rr = 1e-15;
          const double D = 1.0 / Rr;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            // edge direction u_e in the contact frame (normal z, tangents y and -x)
            const double ux = e == 2 ? -mu : (e == 3 ? mu : 0.0), uy = e == 0 ? mu : (e == 1 ? -mu : 0.0);
            const double val = cn + ux * y[0] + uy * y[1];
            if (val < 0.0) {
              fzp -= D * val;   // the edge's force; every edge has a unit normal component
              am |= 1ull << (4 * c + e);
              double J[NV];
#pragma unroll
              for (int j = 0; j < NV; j++) J[j] = ux * Jp[0][j] + uy * Jp[1][j] + Jp[2][j];
#pragma unroll
              for (int i = 0; i < NV; i++) {
                g[i] += D * val * J[i];
#pragma unroll
                for (int j = 0; j <= i; j++) H[cg_tri(i, j)] += D * J[i] * J[j];
              }
            }
          }
        }
        // the group's sums (xor butterfly: the same bits in all 8 lanes)
#pragma unroll
        for (int k = 0; k < NV; k++) g[k] = cg_sum8(g[k]);
#pragma unroll
        for (int k = 0; k < NT; k++) H[k] = cg_sum8(H[k]);
        fz = cg_sum8(fzp);
#pragma unroll
        for (int i = 0; i < NV; i++) {
#pragma unroll
          for (int j = 0; j < NV; j++) g[i] += Mm[i >= j ? cg_tri(i, j) : cg_tri(j, i)] * xk[j];
#pragma unroll
          for (int j = 0; j <= i; j++) H[cg_tri(i, j)] += Mm[cg_tri(i, j)];
        }
        double gn = 0.0;
#pragma unroll
        for (int k = 0; k < NV; k++) gn += g[k] * g[k] / Mm[cg_tri(k, k)];
        if (gn < 1e-22 * (1.0 + ck)) break;
        cg_chol<NV>(H);
        double dx[NV], ng[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) ng[k] = -g[k];
        cg_chol_solve<NV>(H, ng, dx);
        double slope = 0.0;
#pragma unroll
        for (int k = 0; k < NV; k++) slope += g[k] * dx[k];
        double t = 1.0, cn = ck, xn[NV];
        int ls = 0;
        for (; ls < 30; ls++) {   // backtracking; the full step is exact when the active set does not change
#pragma unroll
          for (int k = 0; k < NV; k++) xn[k] = xk[k] + t * dx[k];
          cn = cg_sum8(my_cost(xn)) + quad(xn);
          if (cn <= ck + 1e-4 * t * slope) break;
          t *= 0.5;
        }
#pragma unroll
        for (int k = 0; k < NV; k++) xk[k] = xn[k];
        // A full step that left every lane's active set as it was has landed on the minimiser of the (then exactly quadratic)
        // cost: the next iteration would only form the gradient again to find it zero.  Its normal force comes from the cost
        // evaluation that accepted the step.
        {
          int same = (ls == 0 && cm == am) ? 1 : 0;
          same += __shfl_xor(same, 1); same += __shfl_xor(same, 2); same += __shfl_xor(same, 4);
          if (same == 8) { fz = cg_sum8(cfz); ck = cn; break; }
        }
        if (ck - cn < 1e-16 * (1.0 + fabs(ck))) { ck = cn; break; }
        ck = cn;
      }
      if (threadIdx.x < 64) SF_STAMP(12);
      if (fz > 0.0) {
#pragma unroll
        for (int k = 0; k < NV; k++) qa[k] = r.a0[k] + xk[k];
        // Euler with the hinge damping implicit: (M + h D) qimp = M qacc (MuJoCo's integrator); without hinges qimp = qacc
        if constexpr (NV == 8) {
          double rt[NV], Mh[NT];
#pragma unroll
          for (int i = 0; i < NV; i++) {
            rt[i] = 0.0;
#pragma unroll
            for (int j = 0; j < NV; j++) rt[i] += Mm[i >= j ? cg_tri(i, j) : cg_tri(j, i)] * qa[j];
          }
#pragma unroll
          for (int k = 0; k < NT; k++) Mh[k] = Mm[k];
          Mh[cg_tri(6, 6)] += h * Const::damping;
          Mh[cg_tri(NV - 1, NV - 1)] += h * Const::damping;
          cg_chol<NV>(Mh);
          cg_chol_solve<NV>(Mh, rt, qi);
        } else {
#pragma unroll
          for (int k = 0; k < NV; k++) qi[k] = qa[k];
        }
      }
    }
    if (sub == 0) {
#pragma unroll
      for (int k = 0; k < NV; k++) { G.res[base + grp][k] = qa[k]; G.res[base + grp][8 + k] = qi[k]; }
      G.res[base + grp][16] = fz;
    }
  }
}

template <bool LOAD>
__device__ __forceinline__ void cg_collect(const double* rs, Accel<float>& ex, Accel<float>& im, double* fz_out) {
  const double fz = rs[16];
  if (fz > 0.0) {   // touching geometry but no force (separating): the accelerations stay exactly as they were
    *fz_out = fz;
    ex.lin = mk<float>((float)rs[0], (float)rs[1], (float)rs[2]);
    ex.ang = mk<float>((float)rs[3], (float)rs[4], (float)rs[5]);
    im.lin = mk<float>((float)rs[8], (float)rs[9], (float)rs[10]);
    im.ang = mk<float>((float)rs[11], (float)rs[12], (float)rs[13]);
    if (LOAD) { ex.thdd1 = (float)rs[6]; ex.thdd2 = (float)rs[7]; im.thdd1 = (float)rs[14]; im.thdd2 = (float)rs[15]; }
  }
}

}  // namespace qd
