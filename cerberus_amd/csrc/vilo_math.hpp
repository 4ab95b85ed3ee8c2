// Small fixed-size FP64 math used by the HIP kernels (and by host-side tooling compiled with g++).
// Everything is value-typed and fully unrollable so it lives in VGPRs on gfx950.
// Quaternion / rotation helper semantics follow Cerberus' Utility class
// (src/utils/utility.h:28-81: un-normalised deltaQ, identity positify, Qleft/Qright bottom-right 3x3).
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VD __host__ __device__ __forceinline__
#else
#define VD inline
#endif

namespace vilo {

struct v3 {
  double x, y, z;
};
struct m3 {
  double a[9];  // row-major
  VD double &operator()(int r, int c) { return a[3 * r + c]; }
  VD double operator()(int r, int c) const { return a[3 * r + c]; }
};
struct quat {
  double w, x, y, z;
};

VD v3 mk3(double x, double y, double z) { v3 v; v.x = x; v.y = y; v.z = z; return v; }
VD v3 ld3(const double *p) { return mk3(p[0], p[1], p[2]); }
VD void st3(double *p, const v3 &v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
VD v3 operator+(const v3 &a, const v3 &b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VD v3 operator-(const v3 &a, const v3 &b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VD v3 operator-(const v3 &a) { return mk3(-a.x, -a.y, -a.z); }
VD v3 operator*(const v3 &a, double s) { return mk3(a.x * s, a.y * s, a.z * s); }
VD v3 operator*(double s, const v3 &a) { return a * s; }
VD double dot(const v3 &a, const v3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VD v3 cross(const v3 &a, const v3 &b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
VD double norm(const v3 &a) { return sqrt(dot(a, a)); }
VD double at(const v3 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

VD m3 m3_zero() { m3 m; for (int i = 0; i < 9; ++i) m.a[i] = 0.0; return m; }
VD m3 m3_eye() { m3 m = m3_zero(); m.a[0] = m.a[4] = m.a[8] = 1.0; return m; }
VD m3 operator+(const m3 &a, const m3 &b) { m3 o; for (int i = 0; i < 9; ++i) o.a[i] = a.a[i] + b.a[i]; return o; }
VD m3 operator-(const m3 &a, const m3 &b) { m3 o; for (int i = 0; i < 9; ++i) o.a[i] = a.a[i] - b.a[i]; return o; }
VD m3 operator-(const m3 &a) { m3 o; for (int i = 0; i < 9; ++i) o.a[i] = -a.a[i]; return o; }
VD m3 operator*(const m3 &a, double s) { m3 o; for (int i = 0; i < 9; ++i) o.a[i] = a.a[i] * s; return o; }
VD m3 operator*(const m3 &a, const m3 &b) {
  m3 o;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) o.a[3 * r + c] = a.a[3 * r] * b.a[c] + a.a[3 * r + 1] * b.a[3 + c] + a.a[3 * r + 2] * b.a[6 + c];
  return o;
}
VD v3 operator*(const m3 &a, const v3 &v) {
  return mk3(a.a[0] * v.x + a.a[1] * v.y + a.a[2] * v.z, a.a[3] * v.x + a.a[4] * v.y + a.a[5] * v.z,
             a.a[6] * v.x + a.a[7] * v.y + a.a[8] * v.z);
}
VD m3 tr(const m3 &a) {
  m3 o;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) o.a[3 * r + c] = a.a[3 * c + r];
  return o;
}
// Utility::skewSymmetric (utility.h:43-51)
VD m3 skew(const v3 &q) {
  m3 m;
  m.a[0] = 0; m.a[1] = -q.z; m.a[2] = q.y;
  m.a[3] = q.z; m.a[4] = 0; m.a[5] = -q.x;
  m.a[6] = -q.y; m.a[7] = q.x; m.a[8] = 0;
  return m;
}
VD m3 ld_m3_rowmajor(const double *p) { m3 m; for (int i = 0; i < 9; ++i) m.a[i] = p[i]; return m; }

VD quat mkq(double w, double x, double y, double z) { quat q; q.w = w; q.x = x; q.y = y; q.z = z; return q; }
// pose block [px py pz qx qy qz qw] (estimator.cpp:852-859)
VD quat ldq_pose(const double *pose7) { return mkq(pose7[6], pose7[3], pose7[4], pose7[5]); }
VD quat qmul(const quat &a, const quat &b) {
  return mkq(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w);
}
VD double qn2(const quat &q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
VD quat qinv(const quat &q) { const double n = qn2(q); return mkq(q.w / n, -q.x / n, -q.y / n, -q.z / n); }
VD quat qnormalized(const quat &q) { const double n = sqrt(qn2(q)); return mkq(q.w / n, q.x / n, q.y / n, q.z / n); }
VD v3 qvec(const quat &q) { return mk3(q.x, q.y, q.z); }
// Eigen toRotationMatrix (no normalisation)
VD m3 qR(const quat &q) {
  m3 R;
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R.a[0] = 1 - (tyy + tzz); R.a[1] = txy - twz; R.a[2] = txz + twy;
  R.a[3] = txy + twz; R.a[4] = 1 - (txx + tzz); R.a[5] = tyz - twx;
  R.a[6] = txz - twy; R.a[7] = tyz + twx; R.a[8] = 1 - (txx + tyy);
  return R;
}
// Eigen q * v
VD v3 qrot(const quat &q, const v3 &v) {
  const v3 u = qvec(q);
  const v3 uv = cross(u, v) * 2.0;
  return v + uv * q.w + cross(u, uv);
}
// Utility::deltaQ (utility.h:28-41)
VD quat deltaQ(const v3 &t) { return mkq(1.0, t.x / 2.0, t.y / 2.0, t.z / 2.0); }
VD m3 Qleft33(const quat &q) { return m3_eye() * q.w + skew(qvec(q)); }
VD m3 Qright33(const quat &q) { return m3_eye() * q.w - skew(qvec(q)); }
// (Qleft(a) * Qright(b)).bottomRightCorner<3,3>()
VD m3 QleftQright33(const quat &a, const quat &b) {
  m3 o = Qleft33(a) * Qright33(b);
  const v3 av = qvec(a), bv = qvec(b);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o.a[3 * i + j] -= at(av, i) * at(bv, j);
  return o;
}
// PoseLocalParameterization::Plus (pose_local_parameterization.cpp:12-27)
VD void pose_plus(const double *x, const double *d, double *out) {
  out[0] = x[0] + d[0]; out[1] = x[1] + d[1]; out[2] = x[2] + d[2];
  const quat q = qnormalized(qmul(ldq_pose(x), deltaQ(mk3(d[3], d[4], d[5]))));
  out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
}

// ---- A1 / Go1 leg kinematics (src/legKinematics/A1Kinematics.cpp:43-221) --------------------------
// Written from the chain m = lt sin q1 + lc sin(q1+q2), l = lt cos q1 + lc cos(q1+q2):
//   p = [ox - m, oy + d cos q0 + l sin q0, d sin q0 - l cos q0];  rho_fix = [ox, oy, d, lt], rho_opt = lc.
struct LegKin {
  v3 f;        // fk                                   (A1Kinematics::fk)
  m3 J;        // df/dq, J(r, c) = d f_r / d q_c       (A1Kinematics::jac)
  v3 df_drho;  //                                      (A1Kinematics::dfk_drho)
  m3 dJ[3];    // dJ[k] = dJ/dq_k                      (A1Kinematics::dJ_dq, 9x3 column k reshaped)
  m3 dJ_drho;  //                                      (A1Kinematics::dJ_drho reshaped)
};
VD void leg_fk_jac(const double *q, double lc, const double *rf, v3 &f, m3 &J) {
  double s0, c0, s1, c1, s12, c12;
  s0 = sin(q[0]); c0 = cos(q[0]); s1 = sin(q[1]); c1 = cos(q[1]); s12 = sin(q[1] + q[2]); c12 = cos(q[1] + q[2]);
  const double m = rf[3] * s1 + lc * s12, l = rf[3] * c1 + lc * c12;
  f = mk3(rf[0] - m, rf[1] + rf[2] * c0 + l * s0, rf[2] * s0 - l * c0);
  J.a[0] = 0.0; J.a[1] = -l; J.a[2] = -lc * c12;
  J.a[3] = -rf[2] * s0 + l * c0; J.a[4] = -m * s0; J.a[5] = -lc * s12 * s0;
  J.a[6] = rf[2] * c0 + l * s0; J.a[7] = m * c0; J.a[8] = lc * s12 * c0;
}
VD void leg_kin_full(const double *q, double lc, const double *rf, LegKin &k) {
  double s0, c0, s1, c1, s12, c12;
  s0 = sin(q[0]); c0 = cos(q[0]); s1 = sin(q[1]); c1 = cos(q[1]); s12 = sin(q[1] + q[2]); c12 = cos(q[1] + q[2]);
  const double m = rf[3] * s1 + lc * s12, l = rf[3] * c1 + lc * c12;
  const double m2 = lc * s12, l2 = lc * c12;
  const double d = rf[2];
  k.f = mk3(rf[0] - m, rf[1] + d * c0 + l * s0, d * s0 - l * c0);
  k.J.a[0] = 0.0; k.J.a[1] = -l; k.J.a[2] = -l2;
  k.J.a[3] = -d * s0 + l * c0; k.J.a[4] = -m * s0; k.J.a[5] = -m2 * s0;
  k.J.a[6] = d * c0 + l * s0; k.J.a[7] = m * c0; k.J.a[8] = m2 * c0;
  k.df_drho = mk3(-s12, c12 * s0, -c12 * c0);
  // dJ/dq0
  k.dJ[0].a[0] = 0; k.dJ[0].a[1] = 0; k.dJ[0].a[2] = 0;
  k.dJ[0].a[3] = -d * c0 - l * s0; k.dJ[0].a[4] = -m * c0; k.dJ[0].a[5] = -m2 * c0;
  k.dJ[0].a[6] = -d * s0 + l * c0; k.dJ[0].a[7] = -m * s0; k.dJ[0].a[8] = -m2 * s0;
  // dJ/dq1  (dl/dq1 = -m, dm/dq1 = l, dl2/dq1 = -m2, dm2/dq1 = l2)
  k.dJ[1].a[0] = 0; k.dJ[1].a[1] = m; k.dJ[1].a[2] = m2;
  k.dJ[1].a[3] = -m * c0; k.dJ[1].a[4] = -l * s0; k.dJ[1].a[5] = -l2 * s0;
  k.dJ[1].a[6] = -m * s0; k.dJ[1].a[7] = l * c0; k.dJ[1].a[8] = l2 * c0;
  // dJ/dq2  (dl/dq2 = -m2, dm/dq2 = l2)
  k.dJ[2].a[0] = 0; k.dJ[2].a[1] = m2; k.dJ[2].a[2] = m2;
  k.dJ[2].a[3] = -m2 * c0; k.dJ[2].a[4] = -l2 * s0; k.dJ[2].a[5] = -l2 * s0;
  k.dJ[2].a[6] = -m2 * s0; k.dJ[2].a[7] = l2 * c0; k.dJ[2].a[8] = l2 * c0;
  // dJ/dlc
  k.dJ_drho.a[0] = 0; k.dJ_drho.a[1] = -c12; k.dJ_drho.a[2] = -c12;
  k.dJ_drho.a[3] = c12 * c0; k.dJ_drho.a[4] = -s12 * s0; k.dJ_drho.a[5] = -s12 * s0;
  k.dJ_drho.a[6] = c12 * s0; k.dJ_drho.a[7] = s12 * c0; k.dJ_drho.a[8] = s12 * c0;
}

}  // namespace vilo
