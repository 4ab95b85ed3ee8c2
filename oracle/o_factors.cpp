// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/vilo_oracle.h).
// ceres::CostFunction::Evaluate restatements:
//   IMULegFactor            /root/reference/src/factor/imu_leg_factor.cpp:173-386
//   IMULegIntegrationBase::evaluate  imu_leg_integration_base.cpp:845-898
//   IMUFactor               /root/reference/src/factor/imu_factor.h:28-188 (+ integration_base.h:172-198)
//   Projection*Factor       /root/reference/src/factor/projection{TwoFrameOneCam,TwoFrameTwoCam,OneFrameTwoCam}Factor.cpp
//   PoseLocalParameterization  pose_local_parameterization.cpp:12-35
//   HuberLoss               ceres 1.14 loss_function.cc (rho for s<=a^2: [s,1,0]; else [2a sqrt(s)-a^2, a/sqrt(s), -rho1/(2s)])
//   MarginalizationFactor::Evaluate  marginalization_factor.cpp:347-395
#include <algorithm>
#include <limits>

#include "o_linalg.h"
#include "vilo_oracle.h"

using namespace orc;

// LU with partial pivoting inverse (what Eigen's Matrix::inverse() does for n > 4).
static bool lu_inverse(const double *A, int n, double *inv) {
  std::vector<double> a(A, A + n * n);
  std::vector<int> piv(n);
  for (int i = 0; i < n; ++i) piv[i] = i;
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = std::fabs(a[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(a[i * n + k]) > best) { best = std::fabs(a[i * n + k]); p = i; }
    if (best == 0.0) return false;
    if (p != k) {
      for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[p * n + j]);
      std::swap(piv[k], piv[p]);
    }
    for (int i = k + 1; i < n; ++i) {
      a[i * n + k] /= a[k * n + k];
      const double l = a[i * n + k];
      for (int j = k + 1; j < n; ++j) a[i * n + j] -= l * a[k * n + j];
    }
  }
  // solve A X = I column by column using P A = L U
  for (int c = 0; c < n; ++c) {
    std::vector<double> y(n);
    for (int i = 0; i < n; ++i) {
      double s = (piv[i] == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= a[i * n + k] * y[k];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = y[i];
      for (int k = i + 1; k < n; ++k) s -= a[i * n + k] * inv[k * n + c];
      inv[i * n + c] = s / a[i * n + i];
    }
  }
  return true;
}

// sqrt_info = LLT(cov^-1).matrixL()^T  (upper triangular U with U^T U = cov^-1).
// mode 1 follows the reference literally. mode 0 uses the identity cov = U^-1 U^-T: U^-1 is the unique
// upper-triangular M (positive diagonal) with M M^T = cov, obtained by a Cholesky of the index-reversed
// matrix; U = M^-1. Same mathematical object, no explicit inverse of a badly scaled matrix (its condition number is units: ~ 15 after diagonal equilibration).
extern "C" int orc_sqrt_info(const double *cov, int n, int mode, double *U) {
  if (mode == 1) {
    std::vector<double> inv(n * n);
    if (!lu_inverse(cov, n, inv.data())) return 1;
    // Eigen LLT reads only the lower triangle
    if (!cholesky_lower(inv.data(), n, n)) return 2;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) U[i * n + j] = (j >= i) ? inv[j * n + i] : 0.0;
    return 0;
  }
  std::vector<double> r(n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) r[i * n + j] = cov[(n - 1 - i) * n + (n - 1 - j)];
  if (!cholesky_lower(r.data(), n, n)) return 2;
  // M(i,j) = L(n-1-i, n-1-j): upper triangular
  std::vector<double> M(n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) M[i * n + j] = r[(n - 1 - i) * n + (n - 1 - j)];
  // U = M^-1 by back substitution, column by column
  for (int c = 0; c < n; ++c) {
    for (int i = n - 1; i >= 0; --i) {
      if (i > c) { U[i * n + c] = 0.0; continue; }
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = i + 1; k <= c; ++k) s -= M[i * n + k] * U[k * n + c];
      U[i * n + c] = s / M[i * n + i];
    }
  }
  return 0;
}

template <int N>
static void whiten_and_split(const Mat<N, N> &U, Mat<N, 1> &r, double *residuals) {
  Mat<N, 1> rw = U * r;
  for (int i = 0; i < N; ++i) residuals[i] = rw[i];
}

extern "C" void orc_eval_imu_leg(const orc_config *cfg, const orc_preint *pre, const double *const *par, double *residuals,
                                 double **jacobians) {
  const V3 G = v3(0, 0, cfg->g_norm);
  V3 Pi = v3(par[0]);
  Quat Qi = quat_from_pose(par[0]);
  V3 Vi = v3(par[1]), Bai = v3(par[1] + 3), Bgi = v3(par[1] + 6);
  const double *rhoi = par[2];
  V3 Pj = v3(par[3]);
  Quat Qj = quat_from_pose(par[3]);
  V3 Vj = v3(par[4]), Baj = v3(par[4] + 3), Bgj = v3(par[4] + 6);
  const double *rhoj = par[5];

  Mat<31, 31> Jm;
  std::memcpy(Jm.d, pre->jacobian, sizeof(Jm.d));
  M3 dp_dba = get_block<3, 3>(Jm, 0, 21), dp_dbg = get_block<3, 3>(Jm, 0, 24);
  M3 dq_dbg = get_block<3, 3>(Jm, 3, 24);
  M3 dv_dba = get_block<3, 3>(Jm, 6, 21), dv_dbg = get_block<3, 3>(Jm, 6, 24);
  M3 dep_dbg[4];
  V3 dep_drho[4];
  for (int j = 0; j < 4; ++j) {
    dep_dbg[j] = get_block<3, 3>(Jm, 9 + 3 * j, 24);
    dep_drho[j] = get_block<3, 1>(Jm, 9 + 3 * j, 27 + j);
  }
  const double T = pre->sum_dt;
  Quat delta_q = quat_wxyz(pre->delta_q[3], pre->delta_q[0], pre->delta_q[1], pre->delta_q[2]);
  V3 delta_p = v3(pre->delta_p), delta_v = v3(pre->delta_v);

  // evaluate() imu_leg_integration_base.cpp:845-898
  V3 dba = Bai - v3(pre->lin_ba), dbg = Bgi - v3(pre->lin_bg);
  Quat cq = qmul(delta_q, deltaQ(dq_dbg * dbg));
  V3 cv = delta_v + dv_dba * dba + dv_dbg * dbg;
  V3 cp = delta_p + dp_dba * dba + dp_dbg * dbg;
  Quat Qi_inv = qinv(Qi);
  Mat<31, 1> r;
  V3 rp = qrot(Qi_inv, G * (0.5 * T * T) + Pj - Pi - Vi * T) - cp;
  V3 rq = qvec(qmul(qinv(cq), qmul(Qi_inv, Qj))) * 2.0;
  V3 rv = qrot(Qi_inv, G * T + Vj - Vi) - cv;
  set_block(r, 0, 0, rp);
  set_block(r, 3, 0, rq);
  set_block(r, 6, 0, rv);
  V3 dP = qrot(Qi_inv, Pj - Pi);
  for (int j = 0; j < 4; ++j) {
    double drho = rhoi[j] - pre->lin_rho[j];
    V3 ceps = v3(pre->delta_eps + 3 * j) + dep_dbg[j] * dbg + dep_drho[j] * drho;
    set_block(r, 9 + 3 * j, 0, dP - ceps);
    r[27 + j] = rhoj[j] - rhoi[j];
  }
  set_block(r, 21, 0, Baj - Bai);
  set_block(r, 24, 0, Bgj - Bgi);

  // sqrt_info (imu_leg_factor.cpp:197-198)
  Mat<31, 31> U;
  orc_sqrt_info(pre->covariance, 31, 0, U.d);
  whiten_and_split<31>(U, r, residuals);
  if (!jacobians) return;

  // raw Jacobian, global column layout [pose_i 0..6 | sb_i 7..15 | rho_i 16..19 | pose_j 20..26 | sb_j 27..35 | rho_j 36..39]
  Mat<31, 40> J = Mat<31, 40>::zero();
  const M3 RiT = qR(Qi_inv);
  const M3 I3 = M3::identity();
  // pose_i (:221-252)
  set_block(J, 0, 0, -RiT);
  set_block(J, 0, 3, skew(qrot(Qi_inv, G * (0.5 * T * T) + Pj - Pi - Vi * T)));
  set_block(J, 3, 3, -QleftQright33(qmul(qinv(Qj), Qi), cq));
  set_block(J, 6, 3, skew(qrot(Qi_inv, G * T + Vj - Vi)));
  for (int j = 0; j < 4; ++j) {
    set_block(J, 9 + 3 * j, 0, -RiT);
    set_block(J, 9 + 3 * j, 3, skew(dP));
  }
  // speedbias_i (:254-295)
  set_block(J, 0, 7, -RiT * T);
  set_block(J, 0, 10, -dp_dba);
  set_block(J, 0, 13, -dp_dbg);
  set_block(J, 3, 13, -(Qleft33(qmul(qmul(qinv(Qj), Qi), delta_q)) * dq_dbg));
  set_block(J, 6, 7, -RiT);
  set_block(J, 6, 10, -dv_dba);
  set_block(J, 6, 13, -dv_dbg);
  for (int j = 0; j < 4; ++j) set_block(J, 9 + 3 * j, 13, -dep_dbg[j]);
  set_block(J, 21, 10, -I3);
  set_block(J, 24, 13, -I3);
  // legbias_i (:297-318)
  for (int j = 0; j < 4; ++j) {
    set_block(J, 9 + 3 * j, 16 + j, -dep_drho[j]);
    J(27 + j, 16 + j) = -1.0;
  }
  // pose_j (:320-344)
  set_block(J, 0, 20, RiT);
  set_block(J, 3, 23, Qleft33(qmul(qmul(qinv(cq), Qi_inv), Qj)));
  for (int j = 0; j < 4; ++j) set_block(J, 9 + 3 * j, 20, RiT);
  // speedbias_j (:346-365)
  set_block(J, 6, 27, RiT);
  set_block(J, 21, 30, I3);
  set_block(J, 24, 33, I3);
  // legbias_j (:366-383)
  for (int j = 0; j < 4; ++j) J(27 + j, 36 + j) = 1.0;

  Mat<31, 40> JW = U * J;
  const int off[6] = {0, 7, 16, 20, 27, 36}, sz[6] = {7, 9, 4, 7, 9, 4};
  for (int b = 0; b < 6; ++b)
    if (jacobians[b])
      for (int i = 0; i < 31; ++i)
        for (int c = 0; c < sz[b]; ++c) jacobians[b][i * sz[b] + c] = JW(i, off[b] + c);
}

extern "C" void orc_eval_imu(const orc_config *cfg, const orc_preint_imu *pre, const double *const *par, double *residuals,
                             double **jacobians) {
  const V3 G = v3(0, 0, cfg->g_norm);
  V3 Pi = v3(par[0]);
  Quat Qi = quat_from_pose(par[0]);
  V3 Vi = v3(par[1]), Bai = v3(par[1] + 3), Bgi = v3(par[1] + 6);
  V3 Pj = v3(par[2]);
  Quat Qj = quat_from_pose(par[2]);
  V3 Vj = v3(par[3]), Baj = v3(par[3] + 3), Bgj = v3(par[3] + 6);
  Mat<15, 15> Jm;
  std::memcpy(Jm.d, pre->jacobian, sizeof(Jm.d));
  M3 dp_dba = get_block<3, 3>(Jm, 0, 9), dp_dbg = get_block<3, 3>(Jm, 0, 12);
  M3 dq_dbg = get_block<3, 3>(Jm, 3, 12);
  M3 dv_dba = get_block<3, 3>(Jm, 6, 9), dv_dbg = get_block<3, 3>(Jm, 6, 12);
  const double T = pre->sum_dt;
  Quat delta_q = quat_wxyz(pre->delta_q[3], pre->delta_q[0], pre->delta_q[1], pre->delta_q[2]);
  // integration_base.h:172-198
  V3 dba = Bai - v3(pre->lin_ba), dbg = Bgi - v3(pre->lin_bg);
  Quat cq = qmul(delta_q, deltaQ(dq_dbg * dbg));
  V3 cv = v3(pre->delta_v) + dv_dba * dba + dv_dbg * dbg;
  V3 cp = v3(pre->delta_p) + dp_dba * dba + dp_dbg * dbg;
  Quat Qi_inv = qinv(Qi);
  Mat<15, 1> r;
  set_block(r, 0, 0, qrot(Qi_inv, G * (0.5 * T * T) + Pj - Pi - Vi * T) - cp);
  set_block(r, 3, 0, qvec(qmul(qinv(cq), qmul(Qi_inv, Qj))) * 2.0);
  set_block(r, 6, 0, qrot(Qi_inv, G * T + Vj - Vi) - cv);
  set_block(r, 9, 0, Baj - Bai);
  set_block(r, 12, 0, Bgj - Bgi);
  Mat<15, 15> U;
  orc_sqrt_info(pre->covariance, 15, 0, U.d);
  whiten_and_split<15>(U, r, residuals);
  if (!jacobians) return;
  // columns [pose_i 0..6 | sb_i 7..15 | pose_j 16..22 | sb_j 23..31]
  Mat<15, 32> J = Mat<15, 32>::zero();
  const M3 RiT = qR(Qi_inv);
  const M3 I3 = M3::identity();
  set_block(J, 0, 0, -RiT);
  set_block(J, 0, 3, skew(qrot(Qi_inv, G * (0.5 * T * T) + Pj - Pi - Vi * T)));
  set_block(J, 3, 3, -QleftQright33(qmul(qinv(Qj), Qi), cq));
  set_block(J, 6, 3, skew(qrot(Qi_inv, G * T + Vj - Vi)));
  set_block(J, 0, 7, -RiT * T);
  set_block(J, 0, 10, -dp_dba);
  set_block(J, 0, 13, -dp_dbg);
  set_block(J, 3, 13, -(Qleft33(qmul(qmul(qinv(Qj), Qi), delta_q)) * dq_dbg));
  set_block(J, 6, 7, -RiT);
  set_block(J, 6, 10, -dv_dba);
  set_block(J, 6, 13, -dv_dbg);
  set_block(J, 9, 10, -I3);
  set_block(J, 12, 13, -I3);
  set_block(J, 0, 16, RiT);
  set_block(J, 3, 19, Qleft33(qmul(qmul(qinv(cq), Qi_inv), Qj)));
  set_block(J, 6, 23, RiT);
  set_block(J, 9, 26, I3);
  set_block(J, 12, 29, I3);
  Mat<15, 32> JW = U * J;
  const int off[4] = {0, 7, 16, 23}, sz[4] = {7, 9, 7, 9};
  for (int b = 0; b < 4; ++b)
    if (jacobians[b])
      for (int i = 0; i < 15; ++i)
        for (int c = 0; c < sz[b]; ++c) jacobians[b][i * sz[b] + c] = JW(i, off[b] + c);
}

// ---------------------------------------------------------------------------------------------
// Projection factors. kind 0: TwoFrameOneCam <2,7,7,7,1,1>; 1: TwoFrameTwoCam <2,7,7,7,7,1,1>;
// 2: OneFrameTwoCam <2,7,7,1,1>.
// ---------------------------------------------------------------------------------------------
static void write_2x7(double *dst, const Mat<2, 3> &reduce, const Mat<3, 6> &jaco) {
  Mat<2, 6> m = reduce * jaco;
  for (int r = 0; r < 2; ++r) {
    for (int c = 0; c < 6; ++c) dst[r * 7 + c] = m(r, c);
    dst[r * 7 + 6] = 0.0;
  }
}
static Mat<3, 6> hcat(const M3 &a, const M3 &b) {
  Mat<3, 6> o;
  set_block(o, 0, 0, a);
  set_block(o, 0, 3, b);
  return o;
}

static void eval_proj(const orc_config *cfg, int kind, const double *o, const double *const *par, double *residuals,
                      double **jac) {
  const double sq = cfg->focal_length / 1.5;  // estimator.cpp:124-126
  V3 pts_i = v3(o), pts_j = v3(o + 3);
  V3 vel_i = v3(o[6], o[7], 0), vel_j = v3(o[8], o[9], 0);
  const double td_i = o[10], td_j = o[11];
  V3 Pi = V3::zero(), Pj = V3::zero(), tic, tic2 = V3::zero();
  Quat Qi = quat_wxyz(1, 0, 0, 0), Qj = Qi, qic, qic2 = Qi;
  double inv_dep, td;
  if (kind == 0) {
    Pi = v3(par[0]); Qi = quat_from_pose(par[0]);
    Pj = v3(par[1]); Qj = quat_from_pose(par[1]);
    tic = v3(par[2]); qic = quat_from_pose(par[2]);
    inv_dep = par[3][0]; td = par[4][0];
  } else if (kind == 1) {
    Pi = v3(par[0]); Qi = quat_from_pose(par[0]);
    Pj = v3(par[1]); Qj = quat_from_pose(par[1]);
    tic = v3(par[2]); qic = quat_from_pose(par[2]);
    tic2 = v3(par[3]); qic2 = quat_from_pose(par[3]);
    inv_dep = par[4][0]; td = par[5][0];
  } else {
    tic = v3(par[0]); qic = quat_from_pose(par[0]);
    tic2 = v3(par[1]); qic2 = quat_from_pose(par[1]);
    inv_dep = par[2][0]; td = par[3][0];
  }
  V3 pts_i_td = pts_i - vel_i * (td - td_i);
  V3 pts_j_td = pts_j - vel_j * (td - td_j);
  V3 pts_camera_i;  // pts_i_td / inv_dep_i
  for (int k = 0; k < 3; ++k) pts_camera_i[k] = pts_i_td[k] / inv_dep;
  V3 pts_imu_i = qrot(qic, pts_camera_i) + tic;
  V3 pts_imu_j, pts_camera_j;
  if (kind == 2) {
    pts_imu_j = pts_imu_i;
    pts_camera_j = qrot(qinv(qic2), pts_imu_j - tic2);
  } else {
    V3 pts_w = qrot(Qi, pts_imu_i) + Pi;
    pts_imu_j = qrot(qinv(Qj), pts_w - Pj);
    pts_camera_j = (kind == 0) ? qrot(qinv(qic), pts_imu_j - tic) : qrot(qinv(qic2), pts_imu_j - tic2);
  }
  const double dep_j = pts_camera_j[2];
  residuals[0] = sq * (pts_camera_j[0] / dep_j - pts_j_td[0]);
  residuals[1] = sq * (pts_camera_j[1] / dep_j - pts_j_td[1]);
  if (!jac) return;

  M3 Ri = qR(Qi), Rj = qR(Qj), ric = qR(qic), ric2 = qR(qic2);
  Mat<2, 3> reduce = Mat<2, 3>::zero();
  reduce(0, 0) = 1. / dep_j; reduce(0, 2) = -pts_camera_j[0] / (dep_j * dep_j);
  reduce(1, 1) = 1. / dep_j; reduce(1, 2) = -pts_camera_j[1] / (dep_j * dep_j);
  reduce = reduce * sq;
  const M3 I3 = M3::identity();
  V3 vj2 = vel_j;  // sqrt_info * velocity_j.head(2)
  if (kind == 0) {
    if (jac[0]) write_2x7(jac[0], reduce, hcat(T(ric) * T(Rj), T(ric) * T(Rj) * Ri * (-skew(pts_imu_i))));
    if (jac[1]) write_2x7(jac[1], reduce, hcat(T(ric) * (-T(Rj)), T(ric) * skew(pts_imu_j)));
    if (jac[2]) {
      M3 tmp_r = T(ric) * T(Rj) * Ri * ric;
      M3 right = -(tmp_r * skew(pts_camera_i)) + skew(tmp_r * pts_camera_i) +
                 skew(T(ric) * (T(Rj) * (Ri * tic + Pi - Pj) - tic));
      write_2x7(jac[2], reduce, hcat(T(ric) * (T(Rj) * Ri - I3), right));
    }
    if (jac[3]) {
      Mat<2, 1> jf = reduce * (T(ric) * T(Rj) * Ri * ric * pts_i_td) * -1.0;
      jac[3][0] = jf[0] / (inv_dep * inv_dep); jac[3][1] = jf[1] / (inv_dep * inv_dep);
    }
    if (jac[4]) {
      Mat<2, 1> jt = reduce * (T(ric) * T(Rj) * Ri * ric * vel_i) ;
      jac[4][0] = jt[0] / inv_dep * -1.0 + sq * vj2[0]; jac[4][1] = jt[1] / inv_dep * -1.0 + sq * vj2[1];
    }
  } else if (kind == 1) {
    if (jac[0]) write_2x7(jac[0], reduce, hcat(T(ric2) * T(Rj), T(ric2) * T(Rj) * Ri * (-skew(pts_imu_i))));
    if (jac[1]) write_2x7(jac[1], reduce, hcat(T(ric2) * (-T(Rj)), T(ric2) * skew(pts_imu_j)));
    if (jac[2]) write_2x7(jac[2], reduce, hcat(T(ric2) * T(Rj) * Ri, T(ric2) * T(Rj) * Ri * ric * (-skew(pts_camera_i))));
    if (jac[3]) write_2x7(jac[3], reduce, hcat(-T(ric2), skew(pts_camera_j)));
    if (jac[4]) {
      Mat<2, 1> jf = reduce * (T(ric2) * T(Rj) * Ri * ric * pts_i_td) * -1.0;
      jac[4][0] = jf[0] / (inv_dep * inv_dep); jac[4][1] = jf[1] / (inv_dep * inv_dep);
    }
    if (jac[5]) {
      Mat<2, 1> jt = reduce * (T(ric2) * T(Rj) * Ri * ric * vel_i) ;
      jac[5][0] = jt[0] / inv_dep * -1.0 + sq * vj2[0]; jac[5][1] = jt[1] / inv_dep * -1.0 + sq * vj2[1];
    }
  } else {
    if (jac[0]) write_2x7(jac[0], reduce, hcat(T(ric2), T(ric2) * ric * (-skew(pts_camera_i))));
    if (jac[1]) write_2x7(jac[1], reduce, hcat(-T(ric2), skew(pts_camera_j)));
    if (jac[2]) {
      // NB uses pts_i, not pts_i_td (projectionOneFrameTwoCamFactor.cpp:119)
      Mat<2, 1> jf = reduce * (T(ric2) * ric * pts_i) * -1.0;
      jac[2][0] = jf[0] / (inv_dep * inv_dep); jac[2][1] = jf[1] / (inv_dep * inv_dep);
    }
    if (jac[3]) {
      Mat<2, 1> jt = reduce * (T(ric2) * ric * vel_i) ;
      jac[3][0] = jt[0] / inv_dep * -1.0 + sq * vj2[0]; jac[3][1] = jt[1] / inv_dep * -1.0 + sq * vj2[1];
    }
  }
}

extern "C" void orc_eval_proj2f1c(const orc_config *cfg, const double obs12[12], const double *const *p, double *r, double **j) {
  eval_proj(cfg, 0, obs12, p, r, j);
}
extern "C" void orc_eval_proj2f2c(const orc_config *cfg, const double obs12[12], const double *const *p, double *r, double **j) {
  eval_proj(cfg, 1, obs12, p, r, j);
}
extern "C" void orc_eval_proj1f2c(const orc_config *cfg, const double obs12[12], const double *const *p, double *r, double **j) {
  eval_proj(cfg, 2, obs12, p, r, j);
}

extern "C" void orc_pose_plus(const double x[7], const double delta[6], double out[7]) {
  for (int k = 0; k < 3; ++k) out[k] = x[k] + delta[k];
  Quat q = qnormalized(qmul(quat_from_pose(x), deltaQ(v3(delta + 3))));
  out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
}

extern "C" void orc_huber(double a, double s, double rho[3]) {
  const double b = a * a;
  if (s > b) {
    const double r = std::sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

extern "C" void orc_eval_prior(const orc_prior *pr, const double *const *par, double *residuals, double **jacobians) {
  const int n = pr->n;
  std::vector<double> dx(n, 0.0);
  int xoff = 0;
  for (int b = 0; b < pr->n_blocks; ++b) {
    const int size = pr->block_size[b], idx = pr->block_idx[b];
    const double *x = par[b], *x0 = pr->x0 + xoff;
    if (size != 7) {
      for (int k = 0; k < size; ++k) dx[idx + k] = x[k] - x0[k];
    } else {
      for (int k = 0; k < 3; ++k) dx[idx + k] = x[k] - x0[k];
      Quat dq = qmul(qinv(quat_from_pose(x0)), quat_from_pose(x));
      V3 v = qvec(dq) * 2.0;
      if (!(dq.w >= 0)) v = -v;  // marginalization_factor.cpp:372-375
      for (int k = 0; k < 3; ++k) dx[idx + 3 + k] = v[k];
    }
    xoff += size;
  }
  for (int i = 0; i < n; ++i) {
    double s = pr->r0[i];
    for (int k = 0; k < n; ++k) s += pr->J0[(size_t)i * n + k] * dx[k];
    residuals[i] = s;
  }
  if (!jacobians) return;
  for (int b = 0; b < pr->n_blocks; ++b) {
    if (!jacobians[b]) continue;
    const int size = pr->block_size[b], idx = pr->block_idx[b], ls = (size == 7) ? 6 : size;
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < size; ++c) jacobians[b][i * size + c] = (c < ls) ? pr->J0[(size_t)i * n + idx + c] : 0.0;
  }
}
