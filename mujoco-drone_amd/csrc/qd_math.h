// qd_math.h -- small fixed-size vector helpers shared by the HIP kernels.
// One env lives in one lane, so everything here is scalar-per-lane code that the
// compiler keeps in VGPRs (no runtime-indexed arrays: those would go to scratch).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define QD_HD __host__ __device__ __forceinline__
#else
#define QD_HD inline
#endif

namespace qd {

template <class T>
struct V3 {
  T x, y, z;
};

template <class T> QD_HD V3<T> mk(T x, T y, T z) { V3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <class T> QD_HD V3<T> operator+(V3<T> a, V3<T> b) { return mk<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> QD_HD V3<T> operator-(V3<T> a, V3<T> b) { return mk<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> QD_HD V3<T> operator-(V3<T> a) { return mk<T>(-a.x, -a.y, -a.z); }
template <class T> QD_HD V3<T> operator*(T s, V3<T> a) { return mk<T>(s * a.x, s * a.y, s * a.z); }
template <class T> QD_HD T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> QD_HD V3<T> cross(V3<T> a, V3<T> b) {
  return mk<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// rotation matrix, row-major, body -> world
template <class T>
struct M3 {
  T m00, m01, m02, m10, m11, m12, m20, m21, m22;
};
template <class T> QD_HD V3<T> mul(const M3<T>& R, V3<T> v) {
  return mk<T>(R.m00 * v.x + R.m01 * v.y + R.m02 * v.z, R.m10 * v.x + R.m11 * v.y + R.m12 * v.z,
               R.m20 * v.x + R.m21 * v.y + R.m22 * v.z);
}
template <class T> QD_HD V3<T> mulT(const M3<T>& R, V3<T> v) {
  return mk<T>(R.m00 * v.x + R.m10 * v.y + R.m20 * v.z, R.m01 * v.x + R.m11 * v.y + R.m21 * v.z,
               R.m02 * v.x + R.m12 * v.y + R.m22 * v.z);
}

// (w,x,y,z) unit quaternion -> rotation matrix
template <class T> QD_HD M3<T> quat2mat(T w, T x, T y, T z) {
  M3<T> R;
  const T two = T(2), one = T(1);
  R.m00 = one - two * (y * y + z * z); R.m01 = two * (x * y - w * z);       R.m02 = two * (x * z + w * y);
  R.m10 = two * (x * y + w * z);       R.m11 = one - two * (x * x + z * z); R.m12 = two * (y * z - w * x);
  R.m20 = two * (x * z - w * y);       R.m21 = two * (y * z + w * x);       R.m22 = one - two * (x * x + y * y);
  return R;
}

// ---- scalar math ---------------------------------------------------------------------
// float: the device uses the 1-ulp hardware reciprocal / rsqrt / sqrt (v_rcp_f32, v_rsq_f32,
// v_sqrt_f32) and short polynomial sincos / atan2 / asin instead of the IEEE-exact library
// expansions (10+ instructions per division, 40-60 per trig call): one env runs in one lane
// of a wavefront that has its SIMD to itself, so the step time is the instruction count.
// double: plain libm (host-side test twin, and the float64 algebra core of the load model).
QD_HD float frcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
QD_HD double frcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  // hardware seed (v_rcp_f64, ~1e-8 relative) + two Newton steps instead of the IEEE division expansion
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
QD_HD float frsq(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}
QD_HD double frsq(double x) { return 1.0 / sqrt(x); }
QD_HD float qsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);
#else
  return sqrtf(x);
#endif
}
QD_HD double qsqrt(double x) { return sqrt(x); }
QD_HD float  qabs(float x) { return fabsf(x); }
QD_HD double qabs(double x) { return fabs(x); }
QD_HD float  qfloor(float x) { return floorf(x); }
QD_HD double qfloor(double x) { return floor(x); }
QD_HD float  qcbrt(float x) { return cbrtf(x); }
QD_HD double qcbrt(double x) { return cbrt(x); }
QD_HD float  qlog(float x) { return logf(x); }
QD_HD double qlog(double x) { return log(x); }

// sin and cos of an angle in radians, |a| up to a few thousand (hinge angles, half rotation
// increments).  Cody-Waite reduction by pi/2 in three pieces, degree-7/8 minimax kernels
// (the classic single-precision coefficients); < 1.5 ulp over the reduced range.
QD_HD void qsincos(float a, float* s, float* c) {
  const float k = rintf(a * 0.63661977236758134308f);
  float r = fmaf(-k, 1.5707962512969971f, a);          // pi/2 split: hi
  r = fmaf(-k, 7.5497894158615964e-08f, r);            //             mid
  r = fmaf(-k, 5.3903029534742384e-15f, r);            //             lo
  const float z = r * r;
  float sp = fmaf(z, fmaf(z, fmaf(z, 2.718311493989822e-06f, -1.9841270114e-04f), 8.3333337680e-03f), -1.6666667163e-01f);
  sp = fmaf(r * z, sp, r);
  float cp = fmaf(z, fmaf(z, fmaf(z, 2.443315711809948e-05f, -1.388731625493765e-03f), 4.166664568298827e-02f), -0.5f);
  cp = fmaf(z, cp, 1.0f);
  const int q = (int)k;
  const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}
QD_HD void qsincos(double a, double* s, double* c) { *s = sin(a); *c = cos(a); }

// atan2, branch-free (a lane-divergent branch costs more than the handful of selects): reduce to
// t = min/max in [0,1], one more reduction at tan(pi/8) (Cephes single-precision kernel), then undo.
QD_HD float qatan2(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(fmaxf(ax, ay), 1e-37f), mn = fminf(ax, ay);
  const float t = mn * frcp(mx);
  const bool mid = t > 0.4142135623730950f;
  const float tt = mid ? (t - 1.0f) * frcp(t + 1.0f) : t;
  const float z = tt * tt;
  const float p = fmaf(z, fmaf(z, fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f), 1.99777106478e-1f), -3.33329491539e-1f);
  float r = (mid ? 0.7853981633974483f : 0.0f) + fmaf(p * z, tt, tt);
  r = ay > ax ? 1.5707963267948966f - r : r;
  r = x < 0.f ? 3.14159265358979323846f - r : r;
  return y < 0.f ? -r : r;
}
QD_HD double qatan2(double y, double x) { return atan2(y, x); }
// asin on [-1,1] (Cephes asinf: polynomial for |x| <= 0.5, sqrt identity above)
QD_HD float qasin(float x) {
  const float a = fabsf(x);
  const bool big = a > 0.5f;
  const float z = big ? 0.5f * (1.0f - a) : a * a;
  const float t = big ? qsqrt(z) : a;
  const float p = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 4.2163199048e-2f, 2.4181311049e-2f), 4.5470025998e-2f), 7.4953002686e-2f), 1.6666752422e-1f);
  float r = fmaf(p * z, t, t);
  if (big) r = 1.5707963267948966f - 2.0f * r;
  return x < 0.f ? -r : r;
}
QD_HD double qasin(double x) { return asin(x); }
QD_HD float  qfmod(float a, float b) { return fmodf(a, b); }
QD_HD double qfmod(double a, double b) { return fmod(a, b); }
template <class T> QD_HD T qmax(T a, T b) { return a > b ? a : b; }
template <class T> QD_HD T qmin(T a, T b) { return a < b ? a : b; }
template <class T> QD_HD T qclamp(T x, T lo, T hi) { return qmin(qmax(x, lo), hi); }

// numpy float remainder for a positive constant divisor b (sign follows the divisor): a - floor(a/b)*b.
// The quotient uses the reciprocal (a multiply instead of the 12-instruction IEEE division); if its rounding
// lands on the wrong side of an integer the two corrections below bring the result back into [0, b).
template <class T> QD_HD T npmod(T a, T b) {
  T m = a - qfloor(a * (T(1) / b)) * b;
  if (m < T(0)) m += b;
  if (m >= b) m -= b;
  return m;
}

}  // namespace qd
