// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/vilo_oracle.h).
// A1 leg kinematics and the two preintegrations, restated from
//   /root/reference/src/legKinematics/A1Kinematics.cpp:43-221
//   /root/reference/src/factor/imu_leg_integration_base.cpp:7-470
//   /root/reference/src/factor/integration_base.h:18-170
#include "o_linalg.h"
#include "vilo_oracle.h"

using namespace orc;

// ---------------------------------------------------------------------------------------------
// Kinematics. The reference bodies are MATLAB-generated; here the chain is written out by hand:
//   m  = lt sin(q1) + lc sin(q1+q2)      (fore-aft reach)
//   l  = lt cos(q1) + lc cos(q1+q2)      (leg extension)
//   p  = [ ox - m ;  oy + d cos(q0) + l sin(q0) ;  d sin(q0) - l cos(q0) ]
// which expands to A1Kinematics.cpp:58-66. rho_fix = [ox, oy, d, lt], rho_opt = lc.
// ---------------------------------------------------------------------------------------------
extern "C" void orc_fk(const double q[3], double lc, const double rf[4], double p[3]) {
  const double s0 = std::sin(q[0]), c0 = std::cos(q[0]);
  const double m = rf[3] * std::sin(q[1]) + lc * std::sin(q[1] + q[2]);
  const double l = rf[3] * std::cos(q[1]) + lc * std::cos(q[1] + q[2]);
  p[0] = rf[0] - m;
  p[1] = rf[1] + rf[2] * c0 + l * s0;
  p[2] = rf[2] * s0 - l * c0;
}
// J = dp/dq, column-major 3x3 (A1Kinematics.cpp:69-107)
extern "C" void orc_jac(const double q[3], double lc, const double rf[4], double J[9]) {
  const double s0 = std::sin(q[0]), c0 = std::cos(q[0]);
  const double s12 = std::sin(q[1] + q[2]), c12 = std::cos(q[1] + q[2]);
  const double m = rf[3] * std::sin(q[1]) + lc * s12;
  const double l = rf[3] * std::cos(q[1]) + lc * c12;
  // d/dq0
  J[0] = 0.0;
  J[1] = -rf[2] * s0 + l * c0;
  J[2] = rf[2] * c0 + l * s0;
  // d/dq1: dm/dq1 = l, dl/dq1 = -m
  J[3] = -l;
  J[4] = -m * s0;
  J[5] = m * c0;
  // d/dq2: dm/dq2 = lc c12, dl/dq2 = -lc s12
  J[6] = -lc * c12;
  J[7] = -lc * s12 * s0;
  J[8] = lc * s12 * c0;
}
// dp/dlc (A1Kinematics.cpp:109-120)
extern "C" void orc_dfk_drho(const double q[3], double, const double *, double d[3]) {
  const double s12 = std::sin(q[1] + q[2]), c12 = std::cos(q[1] + q[2]);
  d[0] = -s12;
  d[1] = c12 * std::sin(q[0]);
  d[2] = -c12 * std::cos(q[0]);
}
// d vec(J) / dq, 9x3 column-major: column k = d vec(J)/dq_k, vec column-major (A1Kinematics.cpp:122-193)
extern "C" void orc_dJ_dq(const double q[3], double lc, const double rf[4], double D[27]) {
  const double s0 = std::sin(q[0]), c0 = std::cos(q[0]);
  const double s12 = std::sin(q[1] + q[2]), c12 = std::cos(q[1] + q[2]);
  const double m = rf[3] * std::sin(q[1]) + lc * s12;
  const double l = rf[3] * std::cos(q[1]) + lc * c12;
  const double m2 = lc * s12, l2 = lc * c12;  // q2-only parts
  // J entries (column-major index e = row + 3*col):
  //  e0 = 0            e1 = -d s0 + l c0      e2 = d c0 + l s0
  //  e3 = -l           e4 = -m s0             e5 = m c0
  //  e6 = -l2          e7 = -m2 s0            e8 = m2 c0
  double *k0 = D, *k1 = D + 9, *k2 = D + 18;
  // d/dq0
  k0[0] = 0; k0[1] = -rf[2] * c0 - l * s0; k0[2] = -rf[2] * s0 + l * c0;
  k0[3] = 0; k0[4] = -m * c0; k0[5] = -m * s0;
  k0[6] = 0; k0[7] = -m2 * c0; k0[8] = -m2 * s0;
  // d/dq1 (dl/dq1 = -m, dm/dq1 = l, dl2/dq1 = -m2, dm2/dq1 = l2)
  k1[0] = 0; k1[1] = -m * c0; k1[2] = -m * s0;
  k1[3] = m; k1[4] = -l * s0; k1[5] = l * c0;
  k1[6] = m2; k1[7] = -l2 * s0; k1[8] = l2 * c0;
  // d/dq2 (dl/dq2 = -m2, dm/dq2 = l2, dl2/dq2 = -m2, dm2/dq2 = l2)
  k2[0] = 0; k2[1] = -m2 * c0; k2[2] = -m2 * s0;
  k2[3] = m2; k2[4] = -l2 * s0; k2[5] = l2 * c0;
  k2[6] = m2; k2[7] = -l2 * s0; k2[8] = l2 * c0;
}
// d vec(J)/dlc (A1Kinematics.cpp:195-221)
extern "C" void orc_dJ_drho(const double q[3], double, const double *, double d[9]) {
  const double s0 = std::sin(q[0]), c0 = std::cos(q[0]);
  const double s12 = std::sin(q[1] + q[2]), c12 = std::cos(q[1] + q[2]);
  d[0] = 0; d[1] = c12 * c0; d[2] = c12 * s0;
  d[3] = -c12; d[4] = -s12 * s0; d[5] = s12 * c0;
  d[6] = -c12; d[7] = -s12 * s0; d[8] = s12 * c0;
}

extern "C" void orc_default_config(orc_config *c) {
  // config/a1_config/hardware_a1_vilo_config.yaml:24-48,85-99
  c->acc_n = 0.9; c->acc_n_z = 2.5; c->acc_w = 0.0004; c->gyr_n = 0.05; c->gyr_w = 0.0002;
  c->g_norm = 9.805;
  c->phi_n = 0.00001; c->dphi_n = 0.00001;
  c->rho_c_n = 0.00000001; c->rho_nc_n = 0.00000000001;
  c->v_n_min_xy = 0.001; c->v_n_min_z = 0.005; c->v_n_min = 0.005; c->v_n_max = 900.0;
  c->v_n_force_thres_ratio = 0.8; c->v_n_term1_steep = 10; c->v_n_term2_var_rescale = 1.0e-6;
  c->v_n_term3_distance_rescale = 1.0e-3;
  c->contact_sensor_type = 0; c->pad0 = 0;
  // estimator.cpp:142-163, leg order FL FR RL RR
  const double ox[4] = {0.1805, 0.1805, -0.1805, -0.1805};
  const double oy[4] = {0.047, -0.047, 0.047, -0.047};
  const double d[4] = {0.0838, -0.0838, 0.0838, -0.0838};
  for (int j = 0; j < 4; ++j) {
    c->rho_fix[j][0] = ox[j]; c->rho_fix[j][1] = oy[j]; c->rho_fix[j][2] = d[j]; c->rho_fix[j][3] = 0.21;
  }
  for (int i = 0; i < 3; ++i) c->p_br[i] = 0.0;
  for (int i = 0; i < 9; ++i) c->R_br[i] = (i % 4 == 0) ? 1.0 : 0.0;
  c->focal_length = 460.0;
  c->huber_delta = 1.0;
}

namespace {

// State carried by IMULegIntegrationBase across propagate() calls (imu_leg_integration_base.h:73-134).
struct ILState {
  V3 acc_0, gyr_0;
  double phi_0[12], dphi_0[12], c_0[4];
  V3 delta_p, delta_v;
  Quat delta_q;
  V3 delta_eps[4];
  V3 ba, bg;
  double rho[4];
  double sum_dt;
  Mat<31, 31> jacobian, covariance;
  // contact-force filter state (type 2), imu_leg_integration_base.h:100-108
  double foot_force_min[4], foot_force_max[4];
  double foot_force_window[4][5];
  int foot_force_window_idx[4];
  double foot_force_var[4];
};

M3 mat3_rowmajor(const double *p) {
  M3 m;
  for (int i = 0; i < 9; ++i) m.d[i] = p[i];
  return m;
}
M3 mat3_colmajor(const double *p) {
  M3 m;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) m(r, c) = p[c * 3 + r];
  return m;
}

// (dphi^T kron I3) * D for a 9xN column-major derivative table (imu_leg_integration_base.cpp:266-271):
// result(:,n) = sum_k dphi[k] * D[3k..3k+2, n]
template <int N>
Mat<3, N> kron_apply(const double dphi[3], const double *Dcm) {
  Mat<3, N> o;
  for (int n = 0; n < N; ++n)
    for (int r = 0; r < 3; ++r) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += dphi[k] * Dcm[n * 9 + 3 * k + r];
      o(r, n) = s;
    }
  return o;
}

// One midPointIntegration step (imu_leg_integration_base.cpp:138-470). Updates st in place except
// for the final normalisation/time bookkeeping which propagate() does. Optionally returns F, V.
void il_midpoint(const orc_config &cfg, ILState &st, double dt, const V3 &acc_1, const V3 &gyr_1, const double *phi_1,
                 const double *dphi_1, const double *c_1, Mat<31, 31> *F_out, Mat<31, 46> *V_out) {
  const V3 acc_0 = st.acc_0, gyr_0 = st.gyr_0;
  const Quat dq = st.delta_q;
  // :152-160
  V3 un_acc_0 = qrot(dq, acc_0 - st.ba);
  V3 un_gyr = (gyr_0 + gyr_1) * 0.5 - st.bg;
  Quat rq = qmul(dq, quat_wxyz(1, un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2));
  V3 un_acc_1 = qrot(rq, acc_1 - st.ba);
  V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
  V3 r_dp = st.delta_p + st.delta_v * dt + un_acc * (0.5 * dt * dt);
  V3 r_dv = st.delta_v + un_acc * dt;

  // :163-173
  V3 w0 = gyr_0 - st.bg, w1 = gyr_1 - st.bg;
  M3 Rw0 = skew(w0), Rw1 = skew(w1);
  const M3 R0 = qR(dq), R1 = qR(rq);
  const M3 Rbr = mat3_rowmajor(cfg.R_br);
  const V3 pbr = v3(cfg.p_br);

  // contact flag (:183-229). NB foot_contact_flag is an integer vector in the reference
  // (imu_leg_integration_base.h:84 Vector4i), so the type-2 logistic value is truncated to 0/1.
  int flag[4];
  if (cfg.contact_sensor_type == 0 || cfg.contact_sensor_type == 1) {
    for (int j = 0; j < 4; ++j) flag[j] = c_1[j] >= 0.5 ? 1 : 0;
  } else {
    for (int j = 0; j < 4; ++j) {
      double force_mag = 0.5 * (st.c_0[j] + c_1[j]);
      if (force_mag < st.foot_force_min[j]) st.foot_force_min[j] = 0.9 * st.foot_force_min[j] + 0.1 * force_mag;
      if (force_mag > st.foot_force_max[j]) st.foot_force_max[j] = 0.9 * st.foot_force_max[j] + 0.1 * force_mag;
      st.foot_force_min[j] *= 0.9991;
      st.foot_force_max[j] *= 0.997;
      double thr = st.foot_force_min[j] + cfg.v_n_force_thres_ratio * (st.foot_force_max[j] - st.foot_force_min[j]);
      flag[j] = (int)(1.0 / (1 + std::exp(-cfg.v_n_term1_steep * (force_mag - thr))));
      st.foot_force_window_idx[j]++;
      st.foot_force_window_idx[j] %= 5;
      st.foot_force_window[j][st.foot_force_window_idx[j]] = force_mag;
      double mean = 0;
      for (int k = 0; k < 5; ++k) mean += st.foot_force_window[j][k];
      mean /= 5;
      double ss = 0;
      for (int k = 0; k < 5; ++k) ss += (st.foot_force_window[j][k] - mean) * (st.foot_force_window[j][k] - mean);
      st.foot_force_var[j] = ss / 4;
    }
  }

  // leg odometry velocities (:232-247) and kappa/eta terms (:260-287)
  V3 fi[4], fi1[4], vi[4], vi1[4], r_eps[4], lo_v[4];
  M3 Ji[4], Ji1[4], hi[4], hi1[4];
  V3 gi[4], gi1[4];
  for (int j = 0; j < 4; ++j) {
    const double *rf = cfg.rho_fix[j];
    const double lc = st.rho[j];
    double tmp3[3], tmp9[9], tmp27[27];
    orc_fk(st.phi_0 + 3 * j, lc, rf, tmp3); fi[j] = v3(tmp3);
    orc_fk(phi_1 + 3 * j, lc, rf, tmp3); fi1[j] = v3(tmp3);
    orc_jac(st.phi_0 + 3 * j, lc, rf, tmp9); Ji[j] = mat3_colmajor(tmp9);
    orc_jac(phi_1 + 3 * j, lc, rf, tmp9); Ji1[j] = mat3_colmajor(tmp9);
    V3 dphi0 = v3(st.dphi_0 + 3 * j), dphi1 = v3(dphi_1 + 3 * j);
    vi[j] = -(Rbr * Ji[j] * dphi0) - Rw0 * (pbr + Rbr * fi[j]);
    vi1[j] = -(Rbr * Ji1[j] * dphi1) - Rw1 * (pbr + Rbr * fi1[j]);
    lo_v[j] = (qrot(dq, vi[j]) + qrot(rq, vi1[j])) * 0.5;
    r_eps[j] = st.delta_eps[j] + lo_v[j] * dt;

    orc_dfk_drho(st.phi_0 + 3 * j, lc, rf, tmp3); V3 dfdrho0 = v3(tmp3);
    orc_dfk_drho(phi_1 + 3 * j, lc, rf, tmp3); V3 dfdrho1 = v3(tmp3);
    orc_dJ_drho(st.phi_0 + 3 * j, lc, rf, tmp9);
    gi[j] = -(R0 * (Rbr * kron_apply<1>(st.dphi_0 + 3 * j, tmp9) + Rw0 * Rbr * dfdrho0));
    orc_dJ_drho(phi_1 + 3 * j, lc, rf, tmp9);
    gi1[j] = -(R1 * (Rbr * kron_apply<1>(dphi_1 + 3 * j, tmp9) + Rw1 * Rbr * dfdrho1));
    orc_dJ_dq(st.phi_0 + 3 * j, lc, rf, tmp27);
    hi[j] = R0 * (Rbr * kron_apply<3>(st.dphi_0 + 3 * j, tmp27) + Rw0 * Rbr * Ji[j]);
    orc_dJ_dq(phi_1 + 3 * j, lc, rf, tmp27);
    hi1[j] = R1 * (Rbr * kron_apply<3>(dphi_1 + 3 * j, tmp27) + Rw1 * Rbr * Ji1[j]);
  }

  // velocity / rho noise variances (:288-323)
  double unc[12], rho_unc[4];
  if (cfg.contact_sensor_type == 0 || cfg.contact_sensor_type == 1) {
    for (int j = 0; j < 4; ++j) {
      double n_xy = cfg.v_n_max * (1 - flag[j]) + flag[j] * cfg.v_n_min_xy;
      double n_z = cfg.v_n_max * (1 - flag[j]) + flag[j] * cfg.v_n_min_z;
      unc[3 * j] = n_xy; unc[3 * j + 1] = n_xy; unc[3 * j + 2] = n_z;
    }
  } else {
    for (int j = 0; j < 4; ++j) {
      double n1 = cfg.v_n_max * (1 - flag[j]) + cfg.v_n_min;
      double n2 = cfg.v_n_term2_var_rescale * st.foot_force_var[j];
      V3 tmp = lo_v[j] - st.delta_v;  // uses the *incoming* delta_v (:308)
      for (int k = 0; k < 3; ++k) unc[3 * j + k] = n1 + n2 + cfg.v_n_term3_distance_rescale * tmp[k] * tmp[k];
    }
  }
  int flag_sum = 0;
  for (int j = 0; j < 4; ++j) {
    rho_unc[j] = cfg.rho_c_n * flag[j] + cfg.rho_nc_n;
    flag_sum += flag[j];
  }
  // (:326-351 compute weight_list / sum_delta_epsilon, unused by the residual — SURVEY parity note 5)
  if (flag_sum < 1e-6) {  // :354-358
    for (int j = 0; j < 4; ++j) rho_unc[j] = cfg.rho_nc_n;
    for (int k = 0; k < 12; ++k) unc[k] = 10e10;
  }
  // noise diagonal (:360-374)
  double nd[46];
  {
    const double an2 = cfg.acc_n * cfg.acc_n, anz2 = cfg.acc_n_z * cfg.acc_n_z, gn2 = cfg.gyr_n * cfg.gyr_n;
    const double aw2 = cfg.acc_w * cfg.acc_w, gw2 = cfg.gyr_w * cfg.gyr_w;
    const double pn2 = cfg.phi_n * cfg.phi_n, dpn2 = cfg.dphi_n * cfg.dphi_n;
    double base[30] = {an2, an2, anz2, gn2, gn2, gn2, an2, an2, anz2, gn2, gn2, gn2, aw2, aw2, aw2,
                       gw2, gw2, gw2, pn2, pn2, pn2, pn2, pn2, pn2, dpn2, dpn2, dpn2, dpn2, dpn2, dpn2};
    for (int i = 0; i < 30; ++i) nd[i] = base[i];
    for (int i = 0; i < 12; ++i) nd[30 + i] = unc[i];
    for (int i = 0; i < 4; ++i) nd[42 + i] = rho_unc[i];
  }

  // F and V (:376-465). State order P0 R3 V6 E(9+3j) BA21 BG24 RHO(27+j); noise order
  // Ai0 Gi3 Ai1:6 Gi1:9 BA12 BG15 PHIi18 PHIi1:21 DPHIi24 DPHIi1:27 V(30+3j) NRHO(42+j).
  V3 w_x = (gyr_0 + gyr_1) * 0.5 - st.bg;
  V3 a0 = acc_0 - st.ba, a1 = acc_1 - st.ba;
  M3 Rwx = skew(w_x), Ra0 = skew(a0), Ra1 = skew(a1);
  const M3 I3 = M3::identity();
  M3 kappa_7 = I3 - Rwx * dt;
  Mat<31, 31> F = Mat<31, 31>::zero();
  M3 kappa_1 = (R0 * Ra0) * (-0.5 * dt) + (R1 * Ra1 * kappa_7) * (-0.5 * dt);
  set_block(F, 0, 0, I3);
  set_block(F, 0, 3, kappa_1 * (0.5 * dt));
  set_block(F, 0, 6, I3 * dt);
  set_block(F, 0, 21, (R0 + R1) * (-0.25 * dt * dt));
  set_block(F, 0, 24, (R1 * Ra1) * (0.25 * dt * dt * dt));
  set_block(F, 3, 3, kappa_7);
  set_block(F, 3, 24, I3 * (-1.0 * dt));
  set_block(F, 6, 3, kappa_1);
  set_block(F, 6, 6, I3);
  set_block(F, 6, 21, (R0 + R1) * (-0.5 * dt));
  set_block(F, 6, 24, (R1 * Ra1) * (0.5 * dt * dt));
  for (int j = 0; j < 4; ++j) {
    const int e = 9 + 3 * j;
    set_block(F, e, 3, (R0 * skew(vi[j])) * (-0.5 * dt) - (R1 * skew(vi1[j]) * kappa_7) * (0.5 * dt));
    set_block(F, e, e, I3);
    set_block(F, e, 24,
              (R1 * skew(vi1[j])) * (0.5 * dt * dt) -
                  (R0 * skew(pbr + Rbr * fi[j]) + R1 * skew(pbr + Rbr * fi1[j])) * (0.5 * dt));
    set_block(F, e, 27 + j, (gi[j] + gi1[j]) * (0.5 * dt));
  }
  set_block(F, 21, 21, I3);
  set_block(F, 24, 24, I3);
  for (int j = 0; j < 4; ++j) F(27 + j, 27 + j) = 1.0;

  Mat<31, 46> V = Mat<31, 46>::zero();
  M3 VpG = (R1 * Ra1) * (-0.25 * dt * dt * 0.5 * dt);
  set_block(V, 0, 0, R0 * (0.25 * dt * dt));
  set_block(V, 0, 3, VpG);
  set_block(V, 0, 6, R1 * (0.25 * dt * dt));
  set_block(V, 0, 9, VpG);
  set_block(V, 3, 3, I3 * (0.5 * dt));
  set_block(V, 3, 9, I3 * (0.5 * dt));
  M3 VvG = (R1 * Ra1) * (-0.5 * dt * 0.5 * dt);
  set_block(V, 6, 0, R0 * (0.5 * dt));
  set_block(V, 6, 3, VvG);
  set_block(V, 6, 6, R1 * (0.5 * dt));
  set_block(V, 6, 9, VvG);
  for (int j = 0; j < 4; ++j) {
    const int e = 9 + 3 * j;
    set_block(V, e, 3, (R1 * skew(vi1[j])) * (-0.25 * dt * dt) + (R0 * skew(pbr + Rbr * fi[j])) * (0.5 * dt));
    set_block(V, e, 9, (R1 * skew(vi1[j])) * (-0.25 * dt * dt) + (R1 * skew(pbr + Rbr * fi1[j])) * (0.5 * dt));
    set_block(V, e, 18, hi[j] * (-0.5 * dt));
    set_block(V, e, 21, hi1[j] * (-0.5 * dt));
    set_block(V, e, 24, (R0 * Rbr * Ji[j]) * (-0.5 * dt));
    set_block(V, e, 27, (R1 * Rbr * Ji1[j]) * (-0.5 * dt));
    set_block(V, e, 30 + 3 * j, I3 * (-dt));
  }
  set_block(V, 21, 12, I3 * (-dt));
  set_block(V, 24, 15, I3 * (-dt));
  for (int j = 0; j < 4; ++j) V(27 + j, 42 + j) = -dt;

  // :467-468
  st.jacobian = F * st.jacobian;
  Mat<31, 46> VN;
  for (int i = 0; i < 31; ++i)
    for (int k = 0; k < 46; ++k) VN(i, k) = V(i, k) * nd[k];
  st.covariance = F * st.covariance * T(F) + VN * T(V);

  st.delta_p = r_dp;
  st.delta_q = rq;
  st.delta_v = r_dv;
  for (int j = 0; j < 4; ++j) st.delta_eps[j] = r_eps[j];
  if (F_out) *F_out = F;
  if (V_out) *V_out = V;
}

void il_init(ILState &st, const orc_sample *s0, const double ba[3], const double bg[3], const double rho[4]) {
  st.acc_0 = v3(s0->acc);
  st.gyr_0 = v3(s0->gyr);
  std::memcpy(st.phi_0, s0->phi, sizeof(st.phi_0));
  std::memcpy(st.dphi_0, s0->dphi, sizeof(st.dphi_0));
  std::memcpy(st.c_0, s0->c, sizeof(st.c_0));
  st.delta_p = V3::zero();
  st.delta_v = V3::zero();
  st.delta_q = quat_wxyz(1, 0, 0, 0);
  for (int j = 0; j < 4; ++j) st.delta_eps[j] = V3::zero();
  st.ba = v3(ba);
  st.bg = v3(bg);
  for (int j = 0; j < 4; ++j) st.rho[j] = rho[j];
  st.sum_dt = 0;
  st.jacobian = Mat<31, 31>::identity();
  st.covariance = Mat<31, 31>::zero();
  for (int j = 0; j < 4; ++j) {
    st.foot_force_min[j] = st.foot_force_max[j] = 0;
    st.foot_force_window_idx[j] = 0;
    st.foot_force_var[j] = 0;
    for (int k = 0; k < 5; ++k) st.foot_force_window[j][k] = 0;
  }
}

// propagate() (imu_leg_integration_base.cpp:88-136)
void il_propagate(const orc_config &cfg, ILState &st, const orc_sample &s) {
  il_midpoint(cfg, st, s.dt, v3(s.acc), v3(s.gyr), s.phi, s.dphi, s.c, nullptr, nullptr);
  st.delta_q = qnormalized(st.delta_q);
  st.sum_dt += s.dt;
  st.acc_0 = v3(s.acc);
  st.gyr_0 = v3(s.gyr);
  std::memcpy(st.phi_0, s.phi, sizeof(st.phi_0));
  std::memcpy(st.dphi_0, s.dphi, sizeof(st.dphi_0));
  std::memcpy(st.c_0, s.c, sizeof(st.c_0));
}

}  // namespace

namespace {
void il_export(const ILState &st, orc_preint *out) {
  out->sum_dt = st.sum_dt;
  for (int k = 0; k < 3; ++k) {
    out->delta_p[k] = st.delta_p[k];
    out->delta_v[k] = st.delta_v[k];
    out->lin_ba[k] = st.ba[k];
    out->lin_bg[k] = st.bg[k];
  }
  out->delta_q[0] = st.delta_q.x; out->delta_q[1] = st.delta_q.y; out->delta_q[2] = st.delta_q.z; out->delta_q[3] = st.delta_q.w;
  for (int j = 0; j < 4; ++j) {
    for (int k = 0; k < 3; ++k) out->delta_eps[3 * j + k] = st.delta_eps[j][k];
    out->lin_rho[j] = st.rho[j];
  }
  std::memcpy(out->jacobian, st.jacobian.d, sizeof(out->jacobian));
  std::memcpy(out->covariance, st.covariance.d, sizeof(out->covariance));
}
}  // namespace

extern "C" void orc_preintegrate_imu_leg(const orc_config *cfg, const orc_sample *s0, const orc_sample *samples, int n,
                                         const double ba[3], const double bg[3], const double rho[4], orc_preint *out) {
  ILState st;
  il_init(st, s0, ba, bg, rho);
  for (int i = 0; i < n; ++i) il_propagate(*cfg, st, samples[i]);
  il_export(st, out);
}

// IMULegIntegrationBase::repropagate (imu_leg_integration_base.cpp:62-86) on an object that already integrated its samples: everything
// the constructor initialises is reset EXCEPT the contact-force filter of contact_sensor_type 2 — foot_force_min / max / window / window_idx
// / var are members repropagate() does not touch (:62-86 against :29-41), so the pass starts from the filter state the previous pass
// left and leaves its own behind. ff (in / out): min[4], max[4], var[4], window[4][5], idx[4] as 36 doubles; all zero = a fresh object.
extern "C" void orc_preintegrate_imu_leg_ff(const orc_config *cfg, const orc_sample *s0, const orc_sample *samples, int n,
                                            const double ba[3], const double bg[3], const double rho[4], double ff[36], orc_preint *out) {
  ILState st;
  il_init(st, s0, ba, bg, rho);
  for (int j = 0; j < 4; ++j) {
    st.foot_force_min[j] = ff[j]; st.foot_force_max[j] = ff[4 + j]; st.foot_force_var[j] = ff[8 + j];
    for (int k = 0; k < 5; ++k) st.foot_force_window[j][k] = ff[12 + 5 * j + k];
    st.foot_force_window_idx[j] = (int)ff[32 + j];
  }
  for (int i = 0; i < n; ++i) il_propagate(*cfg, st, samples[i]);
  for (int j = 0; j < 4; ++j) {
    ff[j] = st.foot_force_min[j]; ff[4 + j] = st.foot_force_max[j]; ff[8 + j] = st.foot_force_var[j];
    for (int k = 0; k < 5; ++k) ff[12 + 5 * j + k] = st.foot_force_window[j][k];
    ff[32 + j] = (double)st.foot_force_window_idx[j];
  }
  il_export(st, out);
}

extern "C" void orc_imu_leg_step_FV(const orc_config *cfg, const orc_sample *s0, const orc_sample *s1, const double delta_q[4],
                                    const double ba[3], const double bg[3], const double rho[4], double *F, double *V) {
  ILState st;
  il_init(st, s0, ba, bg, rho);
  st.delta_q = quat_wxyz(delta_q[3], delta_q[0], delta_q[1], delta_q[2]);
  Mat<31, 31> Fm;
  Mat<31, 46> Vm;
  il_midpoint(*cfg, st, s1->dt, v3(s1->acc), v3(s1->gyr), s1->phi, s1->dphi, s1->c, &Fm, &Vm);
  std::memcpy(F, Fm.d, sizeof(Fm.d));
  std::memcpy(V, Vm.d, sizeof(Vm.d));
}

// Classic IMU preintegration (integration_base.h:18-170). Noise: ACC_N on all three accel axes (:31-37).
extern "C" void orc_preintegrate_imu(const orc_config *cfg, const orc_sample *s0, const orc_sample *samples, int n,
                                     const double ba_[3], const double bg_[3], orc_preint_imu *out) {
  V3 acc_0 = v3(s0->acc), gyr_0 = v3(s0->gyr);
  V3 ba = v3(ba_), bg = v3(bg_);
  V3 dp = V3::zero(), dv = V3::zero();
  Quat dq = quat_wxyz(1, 0, 0, 0);
  double sum_dt = 0;
  Mat<15, 15> jac = Mat<15, 15>::identity(), cov = Mat<15, 15>::zero();
  double noise[18];
  for (int i = 0; i < 3; ++i) {
    noise[i] = cfg->acc_n * cfg->acc_n;
    noise[3 + i] = cfg->gyr_n * cfg->gyr_n;
    noise[6 + i] = cfg->acc_n * cfg->acc_n;
    noise[9 + i] = cfg->gyr_n * cfg->gyr_n;
    noise[12 + i] = cfg->acc_w * cfg->acc_w;
    noise[15 + i] = cfg->gyr_w * cfg->gyr_w;
  }
  const M3 I3 = M3::identity();
  for (int s = 0; s < n; ++s) {
    const double dt = samples[s].dt;
    V3 acc_1 = v3(samples[s].acc), gyr_1 = v3(samples[s].gyr);
    // :74-82
    V3 un_acc_0 = qrot(dq, acc_0 - ba);
    V3 un_gyr = (gyr_0 + gyr_1) * 0.5 - bg;
    Quat rq = qmul(dq, quat_wxyz(1, un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2));
    V3 un_acc_1 = qrot(rq, acc_1 - ba);
    V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
    V3 r_dp = dp + dv * dt + un_acc * (0.5 * dt * dt);
    V3 r_dv = dv + un_acc * dt;
    // :87-137
    M3 Rwx = skew(un_gyr), Ra0 = skew(acc_0 - ba), Ra1 = skew(acc_1 - ba);
    M3 R0 = qR(dq), R1 = qR(rq);
    M3 K7 = I3 - Rwx * dt;
    Mat<15, 15> F = Mat<15, 15>::zero();
    set_block(F, 0, 0, I3);
    set_block(F, 0, 3, (R0 * Ra0) * (-0.25 * dt * dt) + (R1 * Ra1 * K7) * (-0.25 * dt * dt));
    set_block(F, 0, 6, I3 * dt);
    set_block(F, 0, 9, (R0 + R1) * (-0.25 * dt * dt));
    set_block(F, 0, 12, (R1 * Ra1) * (-0.25 * dt * dt * -dt));
    set_block(F, 3, 3, K7);
    set_block(F, 3, 12, I3 * (-dt));
    set_block(F, 6, 3, (R0 * Ra0) * (-0.5 * dt) + (R1 * Ra1 * K7) * (-0.5 * dt));
    set_block(F, 6, 6, I3);
    set_block(F, 6, 9, (R0 + R1) * (-0.5 * dt));
    set_block(F, 6, 12, (R1 * Ra1) * (-0.5 * dt * -dt));
    set_block(F, 9, 9, I3);
    set_block(F, 12, 12, I3);
    Mat<15, 18> V = Mat<15, 18>::zero();
    M3 VpG = (R1 * Ra1) * (-0.25 * dt * dt * 0.5 * dt);
    set_block(V, 0, 0, R0 * (0.25 * dt * dt));
    set_block(V, 0, 3, VpG);
    set_block(V, 0, 6, R1 * (0.25 * dt * dt));
    set_block(V, 0, 9, VpG);
    set_block(V, 3, 3, I3 * (0.5 * dt));
    set_block(V, 3, 9, I3 * (0.5 * dt));
    M3 VvG = (R1 * Ra1) * (-0.5 * dt * 0.5 * dt);
    set_block(V, 6, 0, R0 * (0.5 * dt));
    set_block(V, 6, 3, VvG);
    set_block(V, 6, 6, R1 * (0.5 * dt));
    set_block(V, 6, 9, VvG);
    set_block(V, 9, 12, I3 * dt);
    set_block(V, 12, 15, I3 * dt);
    jac = F * jac;
    Mat<15, 18> VN;
    for (int i = 0; i < 15; ++i)
      for (int k = 0; k < 18; ++k) VN(i, k) = V(i, k) * noise[k];
    cov = F * cov * T(F) + VN * T(V);
    // propagate() :142-169
    dp = r_dp;
    dv = r_dv;
    dq = qnormalized(rq);
    sum_dt += dt;
    acc_0 = acc_1;
    gyr_0 = gyr_1;
  }
  out->sum_dt = sum_dt;
  for (int k = 0; k < 3; ++k) {
    out->delta_p[k] = dp[k];
    out->delta_v[k] = dv[k];
    out->lin_ba[k] = ba[k];
    out->lin_bg[k] = bg[k];
  }
  out->delta_q[0] = dq.x; out->delta_q[1] = dq.y; out->delta_q[2] = dq.z; out->delta_q[3] = dq.w;
  std::memcpy(out->jacobian, jac.d, sizeof(out->jacobian));
  std::memcpy(out->covariance, cov.d, sizeof(out->covariance));
}
