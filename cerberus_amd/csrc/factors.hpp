// Factor math of the Cerberus optimisation hot path as inline device functions (gfx950 kernels include
// this; tests/host_check compiles the same functions for the host to compare them with the oracle).
// Jacobians are produced directly in the LOCAL (tangent) parameterisation: pose blocks have 6 columns
// (PoseLocalParameterization::ComputeJacobian is [I6;0], pose_local_parameterization.cpp:29-35).
//
//   proj_factor<K>      Projection{TwoFrameOneCam,TwoFrameTwoCam,OneFrameTwoCam}Factor::Evaluate
//                       (projectionTwoFrameOneCamFactor.cpp:43-150, ...TwoCamFactor.cpp:43-166, ...OneFrameTwoCamFactor.cpp:42-134)
//   huber_correct       ceres::HuberLoss + Corrector (restated at marginalization_factor.cpp:46-77)
//   imu_leg_raw         IMULegIntegrationBase::evaluate + IMULegFactor::Evaluate before whitening
//                       (imu_leg_integration_base.cpp:845-898, imu_leg_factor.cpp:173-386)
//   imu_raw             IntegrationBase::evaluate + IMUFactor::Evaluate before whitening
//                       (integration_base.h:172-198, imu_factor.h:28-188)
//   prior_dx            MarginalizationFactor::Evaluate's dx (marginalization_factor.cpp:357-377)
#pragma once
#include <cstddef>
#include "vilo_math.hpp"

// Per-(chunk, t) Gram slot of the visual factors: packed upper triangle of X^T X, X = the corrected [J | r] rows of the chunk's landmarks
// seen from frame s + t, in 23 columns. d r / d P_j = - d r / d P_i in both two-frame factors (projectionTwoFrameOneCamFactor.cpp:97-107,
// projectionTwoFrameTwoCamFactor.cpp:100-112), so the translation columns are stored once; the order puts everything the left-camera
// factor touches except td into the first 16 columns = one FP64-MFMA tile.
#define GC_T 0             // d r / d P_i (3); the P_j columns are their negatives
#define GC_RI 3            // d r / d theta_i (3)
#define GC_RJ 6            // d r / d theta_j (3)
#define GC_E0 9            // ex0 (6)
#define GC_R 15            // residual
#define GC_E1 16           // ex1 (6)
#define GC_TD 22           // td
#define VILO_GCOLS 23
#define VILO_GRAM 276      // 23 * 24 / 2
#define VILO_GRAM26 351    // consumers walk the 26-column view [pose_s 6 | pose_j 6 | ex0 6 | ex1 6 | td | r] through gram26_index

namespace vilo {

// ---------------------------------------------------------------------------------------------
// Visual factors. obs12 = pts_i(3) pts_j(3) vel_i(2) vel_j(2) td_i td_j. sq = FOCAL_LENGTH / 1.5.
// KIND 0: <2,7,7,7,1,1> (pose_i,pose_j,ex0,lambda,td); 1: <2,7,7,7,7,1,1>; 2: <2,7,7,1,1> (ex0,ex1,lambda,td).
// J_* are 2x6 row-major (12 doubles) / 2 doubles; blocks that do not exist for KIND are left untouched.
// ---------------------------------------------------------------------------------------------
template <int KIND>
VD void proj_factor(const double *o, const double *pose_i, const double *pose_j, const double *ex0, const double *ex1,
                    double inv_dep, double td, double sq, double *r, bool want_jac, double *J_i, double *J_j, double *J_e0,
                    double *J_e1, double *J_l, double *J_td) {
  const v3 pts_i = mk3(o[0], o[1], o[2]), pts_j = mk3(o[3], o[4], o[5]);
  const v3 vel_i = mk3(o[6], o[7], 0.0), vel_j = mk3(o[8], o[9], 0.0);
  const double td_i = o[10], td_j = o[11];
  const v3 tic = ld3(ex0);
  const quat qic = ldq_pose(ex0);
  const v3 pts_i_td = pts_i - vel_i * (td - td_i);
  const v3 pts_j_td = pts_j - vel_j * (td - td_j);
  // one FP64 division each for 1/lambda and 1/z (a v_div sequence is ~20 instructions); products of reciprocals elsewhere
  const double inv_lam = 1.0 / inv_dep;
  const v3 pts_camera_i = mk3(pts_i_td.x * inv_lam, pts_i_td.y * inv_lam, pts_i_td.z * inv_lam);
  const v3 pts_imu_i = qrot(qic, pts_camera_i) + tic;
  v3 pts_imu_j, pts_camera_j;
  v3 Pi = mk3(0, 0, 0), Pj = Pi, tic2 = Pi;
  quat Qi = mkq(1, 0, 0, 0), Qj = Qi, qic2 = Qi;
  if (KIND != 0) { tic2 = ld3(ex1); qic2 = ldq_pose(ex1); }
  if (KIND == 2) {
    pts_imu_j = pts_imu_i;
    pts_camera_j = qrot(qinv(qic2), pts_imu_j - tic2);
  } else {
    Pi = ld3(pose_i); Qi = ldq_pose(pose_i);
    Pj = ld3(pose_j); Qj = ldq_pose(pose_j);
    const v3 pts_w = qrot(Qi, pts_imu_i) + Pi;
    pts_imu_j = qrot(qinv(Qj), pts_w - Pj);
    pts_camera_j = (KIND == 0) ? qrot(qinv(qic), pts_imu_j - tic) : qrot(qinv(qic2), pts_imu_j - tic2);
  }
  const double inv_z = 1.0 / pts_camera_j.z;
  r[0] = sq * (pts_camera_j.x * inv_z - pts_j_td.x);
  r[1] = sq * (pts_camera_j.y * inv_z - pts_j_td.y);
  if (!want_jac) return;

  // reduce = sqrt_info * [1/z 0 -x/z^2; 0 1/z -y/z^2]
  const double r00 = sq * inv_z, r02 = -r00 * pts_camera_j.x * inv_z, r12 = -r00 * pts_camera_j.y * inv_z;
  const double il2 = -(inv_lam * inv_lam), nil = -inv_lam;
  // out(2x3) = reduce * M(3x3): row0 = r00*M.row0 + r02*M.row2 ; row1 = r00*M.row1 + r12*M.row2
  auto red3 = [&](const m3 &M, double *dst, int stride, int c0) {
    for (int c = 0; c < 3; ++c) {
      dst[c0 + c] = r00 * M.a[c] + r02 * M.a[6 + c];
      dst[stride + c0 + c] = r00 * M.a[3 + c] + r12 * M.a[6 + c];
    }
  };
  auto redv = [&](const v3 &v, double *dst) {
    dst[0] = r00 * v.x + r02 * v.z;
    dst[1] = r00 * v.y + r12 * v.z;
  };
  const m3 ric = qR(qic);
  if (KIND == 0) {
    const m3 Ri = qR(Qi), Rj = qR(Qj);
    const m3 A = tr(ric) * tr(Rj);  // ric^T Rj^T
    const m3 ARi = A * Ri;
    red3(A, J_i, 6, 0);
    red3(ARi * (-skew(pts_imu_i)), J_i, 6, 3);
    red3(-A, J_j, 6, 0);
    red3(tr(ric) * skew(pts_imu_j), J_j, 6, 3);
    const m3 tmp_r = ARi * ric;
    red3(tr(ric) * (tr(Rj) * Ri - m3_eye()), J_e0, 6, 0);
    red3(-(tmp_r * skew(pts_camera_i)) + skew(tmp_r * pts_camera_i) + skew(tr(ric) * (tr(Rj) * (Ri * tic + Pi - Pj) - tic)), J_e0,
         6, 3);
    redv((tmp_r * pts_i_td) * il2, J_l);
    redv((tmp_r * vel_i) * nil, J_td);
    J_td[0] += sq * vel_j.x;
    J_td[1] += sq * vel_j.y;
  } else if (KIND == 1) {
    const m3 Ri = qR(Qi), Rj = qR(Qj), ric2 = qR(qic2);
    const m3 A = tr(ric2) * tr(Rj);
    const m3 ARi = A * Ri;
    red3(A, J_i, 6, 0);
    red3(ARi * (-skew(pts_imu_i)), J_i, 6, 3);
    red3(-A, J_j, 6, 0);
    red3(tr(ric2) * skew(pts_imu_j), J_j, 6, 3);
    const m3 ARiric = ARi * ric;
    red3(ARi, J_e0, 6, 0);
    red3(ARiric * (-skew(pts_camera_i)), J_e0, 6, 3);
    red3(-tr(ric2), J_e1, 6, 0);
    red3(skew(pts_camera_j), J_e1, 6, 3);
    redv((ARiric * pts_i_td) * il2, J_l);
    redv((ARiric * vel_i) * nil, J_td);
    J_td[0] += sq * vel_j.x;
    J_td[1] += sq * vel_j.y;
  } else {
    const m3 ric2 = qR(qic2);
    const m3 A = tr(ric2) * ric;
    red3(tr(ric2), J_e0, 6, 0);
    red3(A * (-skew(pts_camera_i)), J_e0, 6, 3);
    red3(-tr(ric2), J_e1, 6, 0);
    red3(skew(pts_camera_j), J_e1, 6, 3);
    // NB pts_i, not pts_i_td (projectionOneFrameTwoCamFactor.cpp:119)
    redv((A * pts_i) * il2, J_l);
    redv((A * vel_i) * nil, J_td);
    J_td[0] += sq * vel_j.x;
    J_td[1] += sq * vel_j.y;
  }
}

// ceres::HuberLoss(a)::Evaluate
VD void huber_rho(double a, double s, double rho[3]) {
  const double b = a * a;
  if (s > b) {
    const double r = sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = a / r;
    if (rho[1] < 2.2250738585072014e-308) rho[1] = 2.2250738585072014e-308;
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}
// Corrector for a 2-row visual block: returns rho(s); scales r in place and gives the Jacobian row
// transform J <- sqrt_rho1 * (J - alpha_sq_norm * r (r^T J)) through (sqrt_rho1, alpha_sq_norm) and the
// UNSCALED residual copy r_raw needed by that formula.
struct Corrector {
  double sqrt_rho1, alpha_sq_norm, residual_scaling, rho0;
};
VD Corrector make_corrector(double a, double sq_norm) {
  double rho[3];
  huber_rho(a, sq_norm, rho);
  Corrector c;
  c.rho0 = rho[0];
  c.sqrt_rho1 = sqrt(rho[1]);
  if (sq_norm == 0.0 || rho[2] <= 0.0) {
    c.residual_scaling = c.sqrt_rho1;
    c.alpha_sq_norm = 0.0;
  } else {
    const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
    const double alpha = 1.0 - sqrt(D);
    c.residual_scaling = c.sqrt_rho1 / (1 - alpha);
    c.alpha_sq_norm = alpha / sq_norm;
  }
  return c;
}
// apply to one Jacobian column pair (j0 = dr0/dx, j1 = dr1/dx) with the uncorrected residual (r0, r1)
VD void correct_col(const Corrector &c, double r0, double r1, double &j0, double &j1) {
  const double rTJ = r0 * j0 + r1 * j1;
  j0 = c.sqrt_rho1 * (j0 - c.alpha_sq_norm * r0 * rTJ);
  j1 = c.sqrt_rho1 * (j1 - c.alpha_sq_norm * r1 * rTJ);
}

// ---------------------------------------------------------------------------------------------
// IMU-leg factor, before whitening. `pre` points at the prepared record (PreintHead, below).
// Residual order P0 R3 V6 E(9+3j) BA21 BG24 RHO(27+j) (parameters.h:135-150).
// Raw Jacobian J[31][ld] in LOCAL columns: [pose_i 0..5 | sb_i 6..14 | rho_i 15..18 | pose_j 19..24 |
// sb_j 25..33 | rho_j 34..37]; only non-zeros are written (caller zero-fills). A.5 of SURVEY.md.
// ---------------------------------------------------------------------------------------------
struct PreintHead {       // the 33 scalars + 93 Jacobian entries a factor needs
  double sum_dt;
  double delta_p[3], delta_q[4] /*xyzw*/, delta_v[3], delta_eps[12];
  double lin_ba[3], lin_bg[3], lin_rho[4];
  double dp_dba[9], dp_dbg[9], dq_dbg[9], dv_dba[9], dv_dbg[9];
  double dep_dbg[4][9];
  double dep_drho[4][3];
};
VD m3 ldm(const double *p) { return ld_m3_rowmajor(p); }

// Gather the 126 values a factor needs from a full IMULegIntegrationBase state
// (imu_leg_integration_base.cpp:852-867; state order P0 R3 V6 E(9+3j) BA21 BG24 RHO(27+j)).
template <class PRE>
VD void fill_preint_head(const PRE &p, PreintHead &h) {
  h.sum_dt = p.sum_dt;
  for (int k = 0; k < 3; ++k) { h.delta_p[k] = p.delta_p[k]; h.delta_v[k] = p.delta_v[k]; h.lin_ba[k] = p.lin_ba[k]; h.lin_bg[k] = p.lin_bg[k]; }
  for (int k = 0; k < 4; ++k) { h.delta_q[k] = p.delta_q[k]; h.lin_rho[k] = p.lin_rho[k]; }
  for (int k = 0; k < 12; ++k) h.delta_eps[k] = p.delta_eps[k];
  for (int e = 0; e < 9; ++e) {
    const int a = e / 3, b = e % 3;
    h.dp_dba[e] = p.jacobian[(0 + a) * 31 + 21 + b];
    h.dp_dbg[e] = p.jacobian[(0 + a) * 31 + 24 + b];
    h.dq_dbg[e] = p.jacobian[(3 + a) * 31 + 24 + b];
    h.dv_dba[e] = p.jacobian[(6 + a) * 31 + 21 + b];
    h.dv_dbg[e] = p.jacobian[(6 + a) * 31 + 24 + b];
    for (int j = 0; j < 4; ++j) h.dep_dbg[j][e] = p.jacobian[(9 + 3 * j + a) * 31 + 24 + b];
  }
  for (int j = 0; j < 4; ++j)
    for (int a = 0; a < 3; ++a) h.dep_drho[j][a] = p.jacobian[(9 + 3 * j + a) * 31 + 27 + j];
}
// Same for an IntegrationBase state (O_P 0 O_R 3 O_V 6 O_BA 9 O_BG 12, parameters.h:118-125).
template <class PRE>
VD void fill_preint_head_imu(const PRE &p, PreintHead &h) {
  h.sum_dt = p.sum_dt;
  for (int k = 0; k < 3; ++k) { h.delta_p[k] = p.delta_p[k]; h.delta_v[k] = p.delta_v[k]; h.lin_ba[k] = p.lin_ba[k]; h.lin_bg[k] = p.lin_bg[k]; }
  for (int k = 0; k < 4; ++k) { h.delta_q[k] = p.delta_q[k]; h.lin_rho[k] = 0.0; }
  for (int k = 0; k < 12; ++k) h.delta_eps[k] = 0.0;
  for (int e = 0; e < 9; ++e) {
    const int a = e / 3, b = e % 3;
    h.dp_dba[e] = p.jacobian[(0 + a) * 15 + 9 + b];
    h.dp_dbg[e] = p.jacobian[(0 + a) * 15 + 12 + b];
    h.dq_dbg[e] = p.jacobian[(3 + a) * 15 + 12 + b];
    h.dv_dba[e] = p.jacobian[(6 + a) * 15 + 9 + b];
    h.dv_dbg[e] = p.jacobian[(6 + a) * 15 + 12 + b];
    for (int j = 0; j < 4; ++j) h.dep_dbg[j][e] = 0.0;
  }
  for (int j = 0; j < 4; ++j)
    for (int a = 0; a < 3; ++a) h.dep_drho[j][a] = 0.0;
}

// Jacobian entry (row, col) goes to J[row * ld + col * cs] (cs = 1: row-major; cs = number of factors, ld = 39 cs: entry-major over a batch).
VD void imu_leg_raw(const PreintHead &P, double g_norm, const double *pose_i, const double *sb_i, const double *lb_i,
                    const double *pose_j, const double *sb_j, const double *lb_j, double *r, bool want_jac, double *J, int ld, int cs = 1) {
  const v3 G = mk3(0, 0, g_norm);
  const v3 Pi = ld3(pose_i), Vi = ld3(sb_i), Bai = ld3(sb_i + 3), Bgi = ld3(sb_i + 6);
  const v3 Pj = ld3(pose_j), Vj = ld3(sb_j), Baj = ld3(sb_j + 3), Bgj = ld3(sb_j + 6);
  const quat Qi = ldq_pose(pose_i), Qj = ldq_pose(pose_j);
  const double T = P.sum_dt;
  const quat delta_q = mkq(P.delta_q[3], P.delta_q[0], P.delta_q[1], P.delta_q[2]);
  const m3 dp_dba = ldm(P.dp_dba), dp_dbg = ldm(P.dp_dbg), dq_dbg = ldm(P.dq_dbg), dv_dba = ldm(P.dv_dba), dv_dbg = ldm(P.dv_dbg);
  const v3 dba = Bai - ld3(P.lin_ba), dbg = Bgi - ld3(P.lin_bg);
  const quat cq = qmul(delta_q, deltaQ(dq_dbg * dbg));
  const v3 cv = ld3(P.delta_v) + dv_dba * dba + dv_dbg * dbg;
  const v3 cp = ld3(P.delta_p) + dp_dba * dba + dp_dbg * dbg;
  const quat Qi_inv = qinv(Qi);
  const v3 a_p = qrot(Qi_inv, G * (0.5 * T * T) + Pj - Pi - Vi * T);
  const v3 a_v = qrot(Qi_inv, G * T + Vj - Vi);
  const v3 dP = qrot(Qi_inv, Pj - Pi);
  st3(r + 0, a_p - cp);
  st3(r + 3, qvec(qmul(qinv(cq), qmul(Qi_inv, Qj))) * 2.0);
  st3(r + 6, a_v - cv);
  for (int j = 0; j < 4; ++j) {
    const double drho = lb_i[j] - P.lin_rho[j];
    const v3 ceps = ld3(P.delta_eps + 3 * j) + ldm(P.dep_dbg[j]) * dbg + ld3(P.dep_drho[j]) * drho;
    st3(r + 9 + 3 * j, dP - ceps);
    r[27 + j] = lb_j[j] - lb_i[j];
  }
  st3(r + 21, Baj - Bai);
  st3(r + 24, Bgj - Bgi);
  if (!want_jac) return;

  auto put = [&](int r0, int c0, const m3 &M) {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) J[(r0 + a) * ld + (c0 + b) * cs] = M.a[3 * a + b];
  };
  const m3 RiT = qR(Qi_inv);
  const m3 nRiT = -RiT;
  const m3 I3 = m3_eye(), nI3 = -I3;
  // pose_i
  put(0, 0, nRiT);
  put(0, 3, skew(a_p));
  put(3, 3, -QleftQright33(qmul(qinv(Qj), Qi), cq));
  put(6, 3, skew(a_v));
  const m3 skdP = skew(dP);
  for (int j = 0; j < 4; ++j) { put(9 + 3 * j, 0, nRiT); put(9 + 3 * j, 3, skdP); }
  // speedbias_i
  put(0, 6, nRiT * T);
  put(0, 9, -dp_dba);
  put(0, 12, -dp_dbg);
  put(3, 12, -(Qleft33(qmul(qmul(qinv(Qj), Qi), delta_q)) * dq_dbg));
  put(6, 6, nRiT);
  put(6, 9, -dv_dba);
  put(6, 12, -dv_dbg);
  for (int j = 0; j < 4; ++j) put(9 + 3 * j, 12, -ldm(P.dep_dbg[j]));
  put(21, 9, nI3);
  put(24, 12, nI3);
  // legbias_i
  for (int j = 0; j < 4; ++j) {
    for (int a = 0; a < 3; ++a) J[(9 + 3 * j + a) * ld + (15 + j) * cs] = -P.dep_drho[j][a];
    J[(27 + j) * ld + (15 + j) * cs] = -1.0;
  }
  // pose_j
  put(0, 19, RiT);
  put(3, 22, Qleft33(qmul(qmul(qinv(cq), Qi_inv), Qj)));
  for (int j = 0; j < 4; ++j) put(9 + 3 * j, 19, RiT);
  // speedbias_j
  put(6, 25, RiT);
  put(21, 28, I3);
  put(24, 31, I3);
  // legbias_j
  for (int j = 0; j < 4; ++j) J[(27 + j) * ld + (34 + j) * cs] = 1.0;
}

// Classic IMU factor before whitening. Residual order P0 R3 V6 BA9 BG12; local columns
// [pose_i 0..5 | sb_i 6..14 | pose_j 15..20 | sb_j 21..29]. Uses the P,R,V,BA,BG parts of PreintHead.
VD void imu_raw(const PreintHead &P, double g_norm, const double *pose_i, const double *sb_i, const double *pose_j,
                const double *sb_j, double *r, bool want_jac, double *J, int ld, int cj = 15, int cs = 1) {
  const v3 G = mk3(0, 0, g_norm);
  const v3 Pi = ld3(pose_i), Vi = ld3(sb_i), Bai = ld3(sb_i + 3), Bgi = ld3(sb_i + 6);
  const v3 Pj = ld3(pose_j), Vj = ld3(sb_j), Baj = ld3(sb_j + 3), Bgj = ld3(sb_j + 6);
  const quat Qi = ldq_pose(pose_i), Qj = ldq_pose(pose_j);
  const double T = P.sum_dt;
  const quat delta_q = mkq(P.delta_q[3], P.delta_q[0], P.delta_q[1], P.delta_q[2]);
  const m3 dp_dba = ldm(P.dp_dba), dp_dbg = ldm(P.dp_dbg), dq_dbg = ldm(P.dq_dbg), dv_dba = ldm(P.dv_dba), dv_dbg = ldm(P.dv_dbg);
  const v3 dba = Bai - ld3(P.lin_ba), dbg = Bgi - ld3(P.lin_bg);
  const quat cq = qmul(delta_q, deltaQ(dq_dbg * dbg));
  const v3 cv = ld3(P.delta_v) + dv_dba * dba + dv_dbg * dbg;
  const v3 cp = ld3(P.delta_p) + dp_dba * dba + dp_dbg * dbg;
  const quat Qi_inv = qinv(Qi);
  const v3 a_p = qrot(Qi_inv, G * (0.5 * T * T) + Pj - Pi - Vi * T);
  const v3 a_v = qrot(Qi_inv, G * T + Vj - Vi);
  st3(r + 0, a_p - cp);
  st3(r + 3, qvec(qmul(qinv(cq), qmul(Qi_inv, Qj))) * 2.0);
  st3(r + 6, a_v - cv);
  st3(r + 9, Baj - Bai);
  st3(r + 12, Bgj - Bgi);
  if (!want_jac) return;
  auto put = [&](int r0, int c0, const m3 &M) {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) J[(r0 + a) * ld + (c0 + b) * cs] = M.a[3 * a + b];
  };
  const m3 RiT = qR(Qi_inv);
  const m3 nRiT = -RiT;
  const m3 I3 = m3_eye(), nI3 = -I3;
  put(0, 0, nRiT);
  put(0, 3, skew(a_p));
  put(3, 3, -QleftQright33(qmul(qinv(Qj), Qi), cq));
  put(6, 3, skew(a_v));
  put(0, 6, nRiT * T);
  put(0, 9, -dp_dba);
  put(0, 12, -dp_dbg);
  put(3, 12, -(Qleft33(qmul(qmul(qinv(Qj), Qi), delta_q)) * dq_dbg));
  put(6, 6, nRiT);
  put(6, 9, -dv_dba);
  put(6, 12, -dv_dbg);
  put(9, 9, nI3);
  put(12, 12, nI3);
  // frame-j blocks start at column cj (15 in the factor's own layout; 19 inside the solver's 38-column IMU-leg layout)
  put(0, cj, RiT);
  put(3, cj + 3, Qleft33(qmul(qmul(qinv(cq), Qi_inv), Qj)));
  put(6, cj + 6, RiT);
  put(9, cj + 9, I3);
  put(12, cj + 12, I3);
}

// ---- The same two factors as a handful of 3 x 3 blocks + a gather table (small batches: imu_fused_body of kernels_solve.hip) ----
// imu_leg_raw / imu_raw write ~300 Jacobian entries one after the other — a single lane's chain of stores. Every entry of [J | r] is
// one of: an entry of seven 3 x 3 matrices that depend on the states (R_i^T, [a_p]x, [a_v]x, [dP]x, the two quaternion-product blocks and
// Qleft(..) dq_dbg), an entry of the preintegration record's head, a residual, or +-1 — times a coefficient out of {1, -1, T, -T}.
// imu_blocks evaluates the seven matrices and the residual with the expressions of the functions above (same values) into a pool;
// imu_gather_table says, for every entry of the 32 x 48 operand image of the whitening, where it comes from, so that each lane of a
// wave fetches the entries it needs for its matrix-core operands itself instead of waiting for one lane to write them all.
#define IB_RIT 0      // R_i^T
#define IB_SKAP 9     // [a_p]x
#define IB_SKAV 18    // [a_v]x
#define IB_SKDP 27    // [dP]x             (leg factor only)
#define IB_QLQR 36    // (Qleft(q_j^-1 q_i) Qright(corrected delta_q)).bottomRightCorner<3,3>()
#define IB_QLDQ 45    // Qleft(q_j^-1 q_i delta_q).bottomRightCorner<3,3>() dq_dbg
#define IB_QL2 54     // Qleft(corrected delta_q^-1 q_i^-1 q_j).bottomRightCorner<3,3>()
#define IB_RES 63     // residual (31, padded to 32)
#define IB_ONE 95     // 1.0
#define IB_N 96
// pool: IB_N doubles. leg: IMULegFactor (31 residuals), else IMUFactor (15). Returns T = sum_dt.
VD double imu_blocks(const PreintHead &P, double g_norm, bool leg, const double *pose_i, const double *sb_i, const double *lb_i,
                     const double *pose_j, const double *sb_j, const double *lb_j, double *pool) {
  const v3 G = mk3(0, 0, g_norm);
  const v3 Pi = ld3(pose_i), Vi = ld3(sb_i), Bai = ld3(sb_i + 3), Bgi = ld3(sb_i + 6);
  const v3 Pj = ld3(pose_j), Vj = ld3(sb_j), Baj = ld3(sb_j + 3), Bgj = ld3(sb_j + 6);
  const quat Qi = ldq_pose(pose_i), Qj = ldq_pose(pose_j);
  const double T = P.sum_dt;
  const quat delta_q = mkq(P.delta_q[3], P.delta_q[0], P.delta_q[1], P.delta_q[2]);
  const m3 dp_dba = ldm(P.dp_dba), dp_dbg = ldm(P.dp_dbg), dq_dbg = ldm(P.dq_dbg), dv_dba = ldm(P.dv_dba), dv_dbg = ldm(P.dv_dbg);
  const v3 dba = Bai - ld3(P.lin_ba), dbg = Bgi - ld3(P.lin_bg);
  const quat cq = qmul(delta_q, deltaQ(dq_dbg * dbg));
  const v3 cv = ld3(P.delta_v) + dv_dba * dba + dv_dbg * dbg;
  const v3 cp = ld3(P.delta_p) + dp_dba * dba + dp_dbg * dbg;
  const quat Qi_inv = qinv(Qi);
  const v3 a_p = qrot(Qi_inv, G * (0.5 * T * T) + Pj - Pi - Vi * T);
  const v3 a_v = qrot(Qi_inv, G * T + Vj - Vi);
  double *r = pool + IB_RES;
  st3(r + 0, a_p - cp);
  st3(r + 3, qvec(qmul(qinv(cq), qmul(Qi_inv, Qj))) * 2.0);
  st3(r + 6, a_v - cv);
  if (leg) {
    const v3 dP = qrot(Qi_inv, Pj - Pi);
    for (int j = 0; j < 4; ++j) {
      const double drho = lb_i[j] - P.lin_rho[j];
      const v3 ceps = ld3(P.delta_eps + 3 * j) + ldm(P.dep_dbg[j]) * dbg + ld3(P.dep_drho[j]) * drho;
      st3(r + 9 + 3 * j, dP - ceps);
      r[27 + j] = lb_j[j] - lb_i[j];
    }
    st3(r + 21, Baj - Bai);
    st3(r + 24, Bgj - Bgi);
    const m3 skdP = skew(dP);
    for (int e = 0; e < 9; ++e) pool[IB_SKDP + e] = skdP.a[e];
  } else {
    st3(r + 9, Baj - Bai);
    st3(r + 12, Bgj - Bgi);
  }
  const m3 RiT = qR(Qi_inv), skap = skew(a_p), skav = skew(a_v);
  const m3 qlqr = QleftQright33(qmul(qinv(Qj), Qi), cq);
  const m3 qldq = Qleft33(qmul(qmul(qinv(Qj), Qi), delta_q)) * dq_dbg;
  const m3 ql2 = Qleft33(qmul(qmul(qinv(cq), Qi_inv), Qj));
  for (int e = 0; e < 9; ++e) {
    pool[IB_RIT + e] = RiT.a[e]; pool[IB_SKAP + e] = skap.a[e]; pool[IB_SKAV + e] = skav.a[e];
    pool[IB_QLQR + e] = qlqr.a[e]; pool[IB_QLDQ + e] = qldq.a[e]; pool[IB_QL2 + e] = ql2.a[e];
  }
  pool[IB_ONE] = 1.0;
  return T;
}

// Where entry (row, col) of the 32 x 48 operand image [J | r] comes from: source = (region << 8 | offset), region 0 = pool (IB_*),
// 1 = the record's head (doubles of PreintHead); coefficient code 0: the entry is a structural zero, 1: +1, 2: -1, 3: +T, 4: -T.
// Packed as (code << 12 | region << 8 | offset). Built at compile time from the same block list as imu_leg_raw / imu_raw write.
struct ImuGatherTable { unsigned short e[32 * 48]; };
constexpr ImuGatherTable imu_gather_table(bool leg) {
  ImuGatherTable t{};
  for (int i = 0; i < 32 * 48; ++i) t.e[i] = 0;
  auto set = [&](int r, int c, unsigned region, unsigned off, unsigned code) { t.e[r * 48 + c] = (unsigned short)((code << 12) | (region << 8) | off); };
  auto put = [&](int r0, int c0, unsigned region, unsigned off, unsigned code) {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) set(r0 + a, c0 + b, region, off + 3 * a + b, code);
  };
  auto diag = [&](int r0, int c0, unsigned code) { for (int a = 0; a < 3; ++a) set(r0 + a, c0 + a, 0, IB_ONE, code); };
  constexpr unsigned POS = 1, NEG = 2, NEGT = 4;
  // the blocks both factors share: rows P 0, R 3, V 6; columns pose_i 0, speed-bias_i 6
  put(0, 0, 0, IB_RIT, NEG); put(0, 3, 0, IB_SKAP, POS); put(3, 3, 0, IB_QLQR, NEG); put(6, 3, 0, IB_SKAV, POS);
  put(0, 6, 0, IB_RIT, NEGT); put(0, 9, 1, 33, NEG); put(0, 12, 1, 42, NEG); put(3, 12, 0, IB_QLDQ, NEG);
  put(6, 6, 0, IB_RIT, NEG); put(6, 9, 1, 60, NEG); put(6, 12, 1, 69, NEG);
  // frame-j blocks start at column 19 in both layouts (the plain IMU factor is embedded in the 38-column layout)
  put(0, 19, 0, IB_RIT, POS); put(3, 22, 0, IB_QL2, POS); put(6, 25, 0, IB_RIT, POS);
  if (leg) {
    for (int j = 0; j < 4; ++j) {
      put(9 + 3 * j, 0, 0, IB_RIT, NEG); put(9 + 3 * j, 3, 0, IB_SKDP, POS);
      put(9 + 3 * j, 12, 1, 78 + 9 * j, NEG);
      for (int a = 0; a < 3; ++a) set(9 + 3 * j + a, 15 + j, 1, 114 + 3 * j + a, NEG);
      set(27 + j, 15 + j, 0, IB_ONE, NEG);
      put(9 + 3 * j, 19, 0, IB_RIT, POS);
      set(27 + j, 34 + j, 0, IB_ONE, POS);
    }
    diag(21, 9, NEG); diag(24, 12, NEG); diag(21, 28, POS); diag(24, 31, POS);
    for (int q = 0; q < 31; ++q) set(q, 38, 0, IB_RES + q, POS);
  } else {
    diag(9, 9, NEG); diag(12, 12, NEG); diag(9, 28, POS); diag(12, 31, POS);
    for (int q = 0; q < 15; ++q) set(q, 38, 0, IB_RES + q, POS);
  }
  return t;
}
static_assert(offsetof(PreintHead, dp_dba) == 33 * 8 && offsetof(PreintHead, dp_dbg) == 42 * 8 && offsetof(PreintHead, dv_dba) == 60 * 8 &&
              offsetof(PreintHead, dv_dbg) == 69 * 8 && offsetof(PreintHead, dep_dbg) == 78 * 8 && offsetof(PreintHead, dep_drho) == 114 * 8,
              "head offsets used by imu_gather_table");

// dx of one kept block of the prior (marginalization_factor.cpp:357-377); size = global size (7 -> 6 local).
VD void prior_dx(const double *x, const double *x0, int size, double *dx) {
  if (size != 7) {
    for (int k = 0; k < size; ++k) dx[k] = x[k] - x0[k];
  } else {
    for (int k = 0; k < 3; ++k) dx[k] = x[k] - x0[k];
    const quat dq = qmul(qinv(ldq_pose(x0)), ldq_pose(x));
    v3 v = qvec(dq) * 2.0;
    if (!(dq.w >= 0)) v = -v;
    st3(dx + 3, v);
  }
}

}  // namespace vilo
