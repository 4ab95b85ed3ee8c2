// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Dependency-free FP64 small-matrix / quaternion helpers for the CPU restatement of the
// Cerberus optimisation hot path. Semantics follow Eigen as used by the reference and
// the reference's own Utility class (/root/reference/src/utils/utility.h:28-81).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace orc {

template <int R, int C>
struct Mat {
  double d[R * C];  // row-major
  double &operator()(int r, int c) { return d[r * C + c]; }
  double operator()(int r, int c) const { return d[r * C + c]; }
  double &operator[](int i) { return d[i]; }
  double operator[](int i) const { return d[i]; }
  static Mat zero() {
    Mat m;
    for (int i = 0; i < R * C; ++i) m.d[i] = 0.0;
    return m;
  }
  static Mat identity() {
    Mat m = zero();
    for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0;
    return m;
  }
};
typedef Mat<3, 1> V3;
typedef Mat<3, 3> M3;

template <int R, int K, int C>
inline Mat<R, C> operator*(const Mat<R, K> &a, const Mat<K, C> &b) {
  Mat<R, C> o;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += a(i, k) * b(k, j);
      o(i, j) = s;
    }
  return o;
}
template <int R, int C>
inline Mat<R, C> operator+(const Mat<R, C> &a, const Mat<R, C> &b) {
  Mat<R, C> o;
  for (int i = 0; i < R * C; ++i) o.d[i] = a.d[i] + b.d[i];
  return o;
}
template <int R, int C>
inline Mat<R, C> operator-(const Mat<R, C> &a, const Mat<R, C> &b) {
  Mat<R, C> o;
  for (int i = 0; i < R * C; ++i) o.d[i] = a.d[i] - b.d[i];
  return o;
}
template <int R, int C>
inline Mat<R, C> operator-(const Mat<R, C> &a) {
  Mat<R, C> o;
  for (int i = 0; i < R * C; ++i) o.d[i] = -a.d[i];
  return o;
}
template <int R, int C>
inline Mat<R, C> operator*(const Mat<R, C> &a, double s) {
  Mat<R, C> o;
  for (int i = 0; i < R * C; ++i) o.d[i] = a.d[i] * s;
  return o;
}
template <int R, int C>
inline Mat<R, C> operator*(double s, const Mat<R, C> &a) {
  return a * s;
}
template <int R, int C>
inline Mat<C, R> T(const Mat<R, C> &a) {
  Mat<C, R> o;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) o(j, i) = a(i, j);
  return o;
}
template <int R, int C, int RR, int CC>
inline void set_block(Mat<R, C> &dst, int r0, int c0, const Mat<RR, CC> &src) {
  for (int i = 0; i < RR; ++i)
    for (int j = 0; j < CC; ++j) dst(r0 + i, c0 + j) = src(i, j);
}
template <int RR, int CC, int R, int C>
inline Mat<RR, CC> get_block(const Mat<R, C> &src, int r0, int c0) {
  Mat<RR, CC> o;
  for (int i = 0; i < RR; ++i)
    for (int j = 0; j < CC; ++j) o(i, j) = src(r0 + i, c0 + j);
  return o;
}
inline V3 v3(double x, double y, double z) {
  V3 v;
  v[0] = x; v[1] = y; v[2] = z;
  return v;
}
inline V3 v3(const double *p) { return v3(p[0], p[1], p[2]); }
inline double dot(const V3 &a, const V3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double norm(const V3 &a) { return std::sqrt(dot(a, a)); }
inline V3 cross(const V3 &a, const V3 &b) {
  return v3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
// Utility::skewSymmetric (utility.h:43-51)
inline M3 skew(const V3 &q) {
  M3 m;
  m(0, 0) = 0; m(0, 1) = -q[2]; m(0, 2) = q[1];
  m(1, 0) = q[2]; m(1, 1) = 0; m(1, 2) = -q[0];
  m(2, 0) = -q[1]; m(2, 1) = q[0]; m(2, 2) = 0;
  return m;
}

// Quaternion, Hamilton convention, components named like Eigen::Quaterniond.
struct Quat {
  double w, x, y, z;
};
inline Quat quat_wxyz(double w, double x, double y, double z) {
  Quat q = {w, x, y, z};
  return q;
}
// pose block layout [px py pz qx qy qz qw] (estimator.cpp:852-859)
inline Quat quat_from_pose(const double *pose7) { return quat_wxyz(pose7[6], pose7[3], pose7[4], pose7[5]); }
inline Quat qmul(const Quat &a, const Quat &b) {
  return quat_wxyz(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
                   a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                   a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                   a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w);
}
inline double qnorm2(const Quat &q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
// Eigen::Quaternion::inverse(): conjugate / squaredNorm
inline Quat qinv(const Quat &q) {
  double n2 = qnorm2(q);
  return quat_wxyz(q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2);
}
inline Quat qnormalized(const Quat &q) {
  double n = std::sqrt(qnorm2(q));
  return quat_wxyz(q.w / n, q.x / n, q.y / n, q.z / n);
}
inline V3 qvec(const Quat &q) { return v3(q.x, q.y, q.z); }
// Eigen::Quaternion::toRotationMatrix() (no normalisation)
inline M3 qR(const Quat &q) {
  M3 R;
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
  R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
  R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy);
  return R;
}
// Eigen q * v (_transformVector): v + w*(2 u x v) + u x (2 u x v)
inline V3 qrot(const Quat &q, const V3 &v) {
  V3 u = qvec(q);
  V3 uv = cross(u, v) * 2.0;
  return v + uv * q.w + cross(u, uv);
}
// Utility::deltaQ (utility.h:28-41): (1, theta/2), NOT normalised
inline Quat deltaQ(const V3 &theta) { return quat_wxyz(1.0, theta[0] / 2.0, theta[1] / 2.0, theta[2] / 2.0); }
// bottom-right 3x3 of Utility::Qleft / Qright (utility.h:63-81); positify is the identity (utility.h:54-61)
inline M3 Qleft33(const Quat &q) { return M3::identity() * q.w + skew(qvec(q)); }
inline M3 Qright33(const Quat &q) { return M3::identity() * q.w - skew(qvec(q)); }
// bottom-right 3x3 of Qleft(a) * Qright(b): -a.vec * b.vec^T + Qleft33(a) Qright33(b)
inline M3 QleftQright33(const Quat &a, const Quat &b) {
  M3 o = Qleft33(a) * Qright33(b);
  V3 av = qvec(a), bv = qvec(b);
  // full 4x4 product, rows 1..3 cols 1..3: [av | L33] * [-bv^T ; R33] = -av bv^T ... careful with signs:
  // Qleft = [[w, -v^T],[v, wI+[v]x]], Qright = [[w, -v^T],[v, wI-[v]x]]
  // (Qleft(a) Qright(b))_{1:3,1:3} = av * (-bv^T) + L33(a) R33(b)
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o(i, j) -= av[i] * bv[j];
  return o;
}

// ---------- dynamic dense matrix (row-major) for the solver / marginalisation ----------
struct DMat {
  int r, c;
  std::vector<double> d;
  DMat() : r(0), c(0) {}
  DMat(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_, 0.0) {}
  double &operator()(int i, int j) { return d[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
};

// In-place lower Cholesky A = L L^T on the leading n x n of a row-major array with stride ld.
// Returns false when a pivot is <= 0 or not finite (Eigen LLT reports NumericalIssue likewise).
inline bool cholesky_lower(double *A, int n, int ld) {
  for (int j = 0; j < n; ++j) {
    double s = A[j * ld + j];
    for (int k = 0; k < j; ++k) s -= A[j * ld + k] * A[j * ld + k];
    if (!(s > 0.0) || !std::isfinite(s)) return false;
    double ljj = std::sqrt(s);
    A[j * ld + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double t = A[i * ld + j];
      for (int k = 0; k < j; ++k) t -= A[i * ld + k] * A[j * ld + k];
      A[i * ld + j] = t / ljj;
    }
  }
  return true;
}
// Solve L L^T x = b in place (L lower from cholesky_lower).
inline void cholesky_solve(const double *L, int n, int ld, double *b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * ld + k] * b[k];
    b[i] = s / L[i * ld + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * ld + i] * b[k];
    b[i] = s / L[i * ld + i];
  }
}

// Cyclic Jacobi symmetric eigen-decomposition: A (n x n, symmetric, row-major) = V diag(w) V^T.
// Eigenvalues sorted ascending (as Eigen::SelfAdjointEigenSolver). V columns are eigenvectors.
void jacobi_eigh(const double *A, int n, double *w, double *V);

}  // namespace orc
