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

// precise math dispatch (float on device, double in the host-side test twin)
QD_HD float  qsqrt(float x) { return sqrtf(x); }
QD_HD double qsqrt(double x) { return sqrt(x); }
QD_HD float  qabs(float x) { return fabsf(x); }
QD_HD double qabs(double x) { return fabs(x); }
QD_HD float  qatan2(float y, float x) { return atan2f(y, x); }
QD_HD double qatan2(double y, double x) { return atan2(y, x); }
QD_HD float  qasin(float x) { return asinf(x); }
QD_HD double qasin(double x) { return asin(x); }
QD_HD float  qfloor(float x) { return floorf(x); }
QD_HD double qfloor(double x) { return floor(x); }
QD_HD float  qfmod(float a, float b) { return fmodf(a, b); }
QD_HD double qfmod(double a, double b) { return fmod(a, b); }
QD_HD float  qcbrt(float x) { return cbrtf(x); }
QD_HD double qcbrt(double x) { return cbrt(x); }
QD_HD float  qlog(float x) { return logf(x); }
QD_HD double qlog(double x) { return log(x); }
QD_HD void qsincos(float a, float* s, float* c) {
#if defined(__HIP_DEVICE_COMPILE__)
  sincosf(a, s, c);
#else
  *s = sinf(a); *c = cosf(a);
#endif
}
QD_HD void qsincos(double a, double* s, double* c) { *s = sin(a); *c = cos(a); }
template <class T> QD_HD T qmax(T a, T b) { return a > b ? a : b; }
template <class T> QD_HD T qmin(T a, T b) { return a < b ? a : b; }
template <class T> QD_HD T qclamp(T x, T lo, T hi) { return qmin(qmax(x, lo), hi); }

// numpy float remainder: sign follows the divisor (b > 0 here)
template <class T> QD_HD T npmod(T a, T b) {
  T m = qfmod(a, b);
  if (m < T(0)) m += b;
  return m;
}

}  // namespace qd
