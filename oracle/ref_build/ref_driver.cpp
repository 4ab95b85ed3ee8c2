// ref_driver.cpp — TEST INFRASTRUCTURE. C entry points around the reference's OWN factor classes, which are compiled
// unmodified from /root/reference/src (see Makefile) against the header shim in shim/. The entry points take the same
// PODs as the oracle (oracle/vilo_oracle.h) so the tests can run oracle and reference side by side on the same inputs.
//
// What is the reference's code here: A1Kinematics, IMULegIntegrationBase, IntegrationBase, IMULegFactor, IMUFactor,
// the three Projection*Factor classes, PoseLocalParameterization, ResidualBlockInfo / MarginalizationInfo /
// MarginalizationFactor. What is NOT: the globals of parameters.cpp (set here from orc_config because parameters.cpp
// needs OpenCV's FileStorage) and the residual-block enumeration of Estimator::optimization (estimator.cpp:1247-1455
// cannot be compiled without the whole ROS/OpenCV estimator; ref_marginalize below re-enumerates the blocks in the same
// order and hands them to the reference's MarginalizationInfo).
#include <cstring>
#include <ctime>
#include <unordered_map>
#include <vector>

#include "factor/imu_factor.h"
#include "factor/imu_leg_factor.h"
#include "factor/marginalization_factor.h"
#include "factor/pose_local_parameterization.h"
#include "factor/projectionOneFrameTwoCamFactor.h"
#include "factor/projectionTwoFrameOneCamFactor.h"
#include "factor/projectionTwoFrameTwoCamFactor.h"
#include "legKinematics/A1Kinematics.h"

#include "../vilo_oracle.h"

// ---- globals declared in utils/parameters.h that the compiled reference objects link against
double ACC_N, ACC_N_Z, ACC_W, GYR_N, GYR_W;
Eigen::Vector3d G{0.0, 0.0, 9.8};
int CONTACT_SENSOR_TYPE;
double PHI_N, DPHI_N, RHO_C_N, RHO_NC_N;
double V_N_MIN_XY, V_N_MIN_Z, V_N_MIN, V_N_MAX, V_N_FORCE_THRES_RATIO, V_N_TERM1_STEEP, V_N_TERM2_VAR_RESCALE, V_N_TERM3_DISTANCE_RESCALE;

namespace {
std::vector<Eigen::VectorXd> g_rho_fix(4);
Eigen::Vector3d g_p_br;
Eigen::Matrix3d g_R_br;
double g_huber = 1.0;

Eigen::Vector3d v3(const double *p) { return Eigen::Vector3d(p[0], p[1], p[2]); }
template <int N> Eigen::Matrix<double, N, 1> vn(const double *p) { Eigen::Matrix<double, N, 1> v; for (int i = 0; i < N; ++i) v(i) = p[i]; return v; }

void apply_config(const orc_config *c) {
  ACC_N = c->acc_n; ACC_N_Z = c->acc_n_z; ACC_W = c->acc_w; GYR_N = c->gyr_n; GYR_W = c->gyr_w;
  G = Eigen::Vector3d(0.0, 0.0, c->g_norm);
  PHI_N = c->phi_n; DPHI_N = c->dphi_n; RHO_C_N = c->rho_c_n; RHO_NC_N = c->rho_nc_n;
  V_N_MIN_XY = c->v_n_min_xy; V_N_MIN_Z = c->v_n_min_z; V_N_MIN = c->v_n_min; V_N_MAX = c->v_n_max;
  V_N_FORCE_THRES_RATIO = c->v_n_force_thres_ratio; V_N_TERM1_STEEP = c->v_n_term1_steep;
  V_N_TERM2_VAR_RESCALE = c->v_n_term2_var_rescale; V_N_TERM3_DISTANCE_RESCALE = c->v_n_term3_distance_rescale;
  CONTACT_SENSOR_TYPE = c->contact_sensor_type;
  for (int j = 0; j < 4; ++j) { g_rho_fix[j].resize(4); for (int k = 0; k < 4; ++k) g_rho_fix[j](k) = c->rho_fix[j][k]; }
  g_p_br = v3(c->p_br);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) g_R_br(i, j) = c->R_br[3 * i + j];
  // estimator.cpp:124-126
  ProjectionTwoFrameOneCamFactor::sqrt_info = c->focal_length / 1.5 * Eigen::Matrix2d::Identity();
  ProjectionTwoFrameTwoCamFactor::sqrt_info = c->focal_length / 1.5 * Eigen::Matrix2d::Identity();
  ProjectionOneFrameTwoCamFactor::sqrt_info = c->focal_length / 1.5 * Eigen::Matrix2d::Identity();
  g_huber = c->huber_delta;
}

IMULegIntegrationBase *new_il(const orc_sample *s0, const double ba[3], const double bg[3], const double rho[4]) {
  return new IMULegIntegrationBase(v3(s0->acc), v3(s0->gyr), vn<12>(s0->phi), vn<12>(s0->dphi), vn<4>(s0->c), v3(ba), v3(bg), vn<4>(rho),
                                   g_rho_fix, g_p_br, g_R_br);
}

// An integration object whose public result fields are overwritten from a stored record.
IMULegIntegrationBase *il_from_record(const orc_preint *p) {
  orc_sample z; std::memset(&z, 0, sizeof z);
  IMULegIntegrationBase *il = new_il(&z, p->lin_ba, p->lin_bg, p->lin_rho);
  il->sum_dt = p->sum_dt;
  il->delta_p = v3(p->delta_p);
  il->delta_q = Eigen::Quaterniond(p->delta_q[3], p->delta_q[0], p->delta_q[1], p->delta_q[2]);
  il->delta_v = v3(p->delta_v);
  il->delta_epsilon.resize(4);
  for (int j = 0; j < 4; ++j) il->delta_epsilon[j] = v3(p->delta_eps + 3 * j);
  for (int i = 0; i < 31; ++i) for (int j = 0; j < 31; ++j) { il->jacobian(i, j) = p->jacobian[31 * i + j]; il->covariance(i, j) = p->covariance[31 * i + j]; }
  return il;
}
IntegrationBase *imu_from_record(const orc_preint_imu *p) {
  IntegrationBase *ib = new IntegrationBase(Eigen::Vector3d::Zero(), Eigen::Vector3d::Zero(), v3(p->lin_ba), v3(p->lin_bg));
  ib->sum_dt = p->sum_dt;
  ib->delta_p = v3(p->delta_p);
  ib->delta_q = Eigen::Quaterniond(p->delta_q[3], p->delta_q[0], p->delta_q[1], p->delta_q[2]);
  ib->delta_v = v3(p->delta_v);
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { ib->jacobian(i, j) = p->jacobian[15 * i + j]; ib->covariance(i, j) = p->covariance[15 * i + j]; }
  return ib;
}
}  // namespace

extern "C" {

void ref_set_config(const orc_config *c) { apply_config(c); }

// ---- A1Kinematics (outputs column-major like Eigen's .data())
void ref_fk(const double q[3], double lc, const double rho_fix[4], double p[3]) {
  A1Kinematics k; Eigen::VectorXd ro(1); ro(0) = lc;
  Eigen::Vector3d o = k.fk(v3(q), ro, Eigen::VectorXd(vn<4>(rho_fix)));
  for (int i = 0; i < 3; ++i) p[i] = o(i);
}
void ref_jac(const double q[3], double lc, const double rho_fix[4], double J[9]) {
  A1Kinematics k; Eigen::VectorXd ro(1); ro(0) = lc;
  Eigen::Matrix3d o = k.jac(v3(q), ro, Eigen::VectorXd(vn<4>(rho_fix)));
  std::memcpy(J, o.data(), sizeof(double) * 9);
}
void ref_dfk_drho(const double q[3], double lc, const double rho_fix[4], double d[3]) {
  A1Kinematics k; Eigen::VectorXd ro(1); ro(0) = lc;
  Eigen::Matrix<double, 3, 1> o = k.dfk_drho(v3(q), ro, Eigen::VectorXd(vn<4>(rho_fix)));
  std::memcpy(d, o.data(), sizeof(double) * 3);
}
void ref_dJ_dq(const double q[3], double lc, const double rho_fix[4], double d[27]) {
  A1Kinematics k; Eigen::VectorXd ro(1); ro(0) = lc;
  Eigen::Matrix<double, 9, 3> o = k.dJ_dq(v3(q), ro, Eigen::VectorXd(vn<4>(rho_fix)));
  std::memcpy(d, o.data(), sizeof(double) * 27);
}
void ref_dJ_drho(const double q[3], double lc, const double rho_fix[4], double d[9]) {
  A1Kinematics k; Eigen::VectorXd ro(1); ro(0) = lc;
  Eigen::Matrix<double, 9, 1> o = k.dJ_drho(v3(q), ro, Eigen::VectorXd(vn<4>(rho_fix)));
  std::memcpy(d, o.data(), sizeof(double) * 9);
}

// ---- preintegration: constructor + push_back per sample
static void export_il(const IMULegIntegrationBase *il, orc_preint *out) {
  out->sum_dt = il->sum_dt;
  for (int i = 0; i < 3; ++i) { out->delta_p[i] = il->delta_p(i); out->delta_v[i] = il->delta_v(i); out->lin_ba[i] = il->linearized_ba(i); out->lin_bg[i] = il->linearized_bg(i); }
  out->delta_q[0] = il->delta_q.x(); out->delta_q[1] = il->delta_q.y(); out->delta_q[2] = il->delta_q.z(); out->delta_q[3] = il->delta_q.w();
  for (int j = 0; j < 4; ++j) { out->lin_rho[j] = il->linearized_rho(j); for (int i = 0; i < 3; ++i) out->delta_eps[3 * j + i] = il->delta_epsilon[j](i); }
  for (int i = 0; i < 31; ++i) for (int j = 0; j < 31; ++j) { out->jacobian[31 * i + j] = il->jacobian(i, j); out->covariance[31 * i + j] = il->covariance(i, j); }
}
void ref_preintegrate_imu_leg(const orc_config *cfg, const orc_sample *s0, const orc_sample *samples, int n, const double ba[3],
                              const double bg[3], const double rho[4], orc_preint *out) {
  apply_config(cfg);
  IMULegIntegrationBase *il = new_il(s0, ba, bg, rho);
  for (int k = 0; k < n; ++k) {
    const orc_sample &s = samples[k];
    il->push_back(s.dt, v3(s.acc), v3(s.gyr), vn<12>(s.phi), vn<12>(s.dphi), vn<4>(s.c));
  }
  export_il(il, out);
  delete il;
}
// the same object integrated once (constructor + push_back at lin0 = ba bg rho), then IMULegIntegrationBase::repropagate() n_rep times at
// lins[r] = ba(3) bg(3) rho(4); out[0] = the state after the original integration, out[1 + r] = after repropagate number r
void ref_repropagate_imu_leg(const orc_config *cfg, const orc_sample *s0, const orc_sample *samples, int n, const double lin0[10], int n_rep,
                             const double *lins, orc_preint *out) {
  apply_config(cfg);
  IMULegIntegrationBase *il = new_il(s0, lin0, lin0 + 3, lin0 + 6);
  for (int k = 0; k < n; ++k) {
    const orc_sample &s = samples[k];
    il->push_back(s.dt, v3(s.acc), v3(s.gyr), vn<12>(s.phi), vn<12>(s.dphi), vn<4>(s.c));
  }
  export_il(il, out);
  for (int r = 0; r < n_rep; ++r) {
    const double *l = lins + 10 * r;
    il->repropagate(v3(l), v3(l + 3), vn<4>(l + 6));
    export_il(il, out + 1 + r);
  }
  delete il;
}
void ref_preintegrate_imu(const orc_config *cfg, const orc_sample *s0, const orc_sample *samples, int n, const double ba[3], const double bg[3],
                          orc_preint_imu *out) {
  apply_config(cfg);
  IntegrationBase ib(v3(s0->acc), v3(s0->gyr), v3(ba), v3(bg));
  for (int k = 0; k < n; ++k) ib.push_back(samples[k].dt, v3(samples[k].acc), v3(samples[k].gyr));
  out->sum_dt = ib.sum_dt;
  for (int i = 0; i < 3; ++i) { out->delta_p[i] = ib.delta_p(i); out->delta_v[i] = ib.delta_v(i); out->lin_ba[i] = ib.linearized_ba(i); out->lin_bg[i] = ib.linearized_bg(i); }
  out->delta_q[0] = ib.delta_q.x(); out->delta_q[1] = ib.delta_q.y(); out->delta_q[2] = ib.delta_q.z(); out->delta_q[3] = ib.delta_q.w();
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { out->jacobian[15 * i + j] = ib.jacobian(i, j); out->covariance[15 * i + j] = ib.covariance(i, j); }
}

// ---- ceres::CostFunction::Evaluate of the reference's factor classes
void ref_eval_imu_leg(const orc_config *cfg, const orc_preint *pre, const double *const *parameters, double *residuals, double **jacobians) {
  apply_config(cfg);
  IMULegIntegrationBase *il = il_from_record(pre);
  IMULegFactor f(il);
  f.Evaluate(parameters, residuals, jacobians);
  delete il;
}
void ref_eval_imu(const orc_config *cfg, const orc_preint_imu *pre, const double *const *parameters, double *residuals, double **jacobians) {
  apply_config(cfg);
  IntegrationBase *ib = imu_from_record(pre);
  IMUFactor f(ib);
  f.Evaluate(parameters, residuals, jacobians);
  delete ib;
}
void ref_eval_proj2f1c(const orc_config *cfg, const double o[12], const double *const *parameters, double *residuals, double **jacobians) {
  apply_config(cfg);
  ProjectionTwoFrameOneCamFactor f(v3(o), v3(o + 3), Eigen::Vector2d(o[6], o[7]), Eigen::Vector2d(o[8], o[9]), o[10], o[11]);
  f.Evaluate(parameters, residuals, jacobians);
}
void ref_eval_proj2f2c(const orc_config *cfg, const double o[12], const double *const *parameters, double *residuals, double **jacobians) {
  apply_config(cfg);
  ProjectionTwoFrameTwoCamFactor f(v3(o), v3(o + 3), Eigen::Vector2d(o[6], o[7]), Eigen::Vector2d(o[8], o[9]), o[10], o[11]);
  f.Evaluate(parameters, residuals, jacobians);
}
void ref_eval_proj1f2c(const orc_config *cfg, const double o[12], const double *const *parameters, double *residuals, double **jacobians) {
  apply_config(cfg);
  ProjectionOneFrameTwoCamFactor f(v3(o), v3(o + 3), Eigen::Vector2d(o[6], o[7]), Eigen::Vector2d(o[8], o[9]), o[10], o[11]);
  f.Evaluate(parameters, residuals, jacobians);
}
void ref_pose_plus(const double x[7], const double delta[6], double out[7]) {
  PoseLocalParameterization p;
  static_cast<const ceres::LocalParameterization &>(p).Plus(x, delta, out);   // Plus is private in the subclass, public in the interface
}

// ---- prior: a MarginalizationInfo whose result fields are filled from a stored record (m = 0)
static MarginalizationInfo *info_from_record(const orc_prior *p, std::vector<double *> *blocks_out, const std::unordered_map<int, double *> *addr_of_id) {
  MarginalizationInfo *info = new MarginalizationInfo();
  info->m = 0; info->n = p->n; info->valid = p->valid != 0;
  int off = 0;
  for (int k = 0; k < p->n_blocks; ++k) {
    info->keep_block_size.push_back(p->block_size[k]);
    info->keep_block_idx.push_back(p->block_idx[k]);
    double *d = new double[p->block_size[k]];
    std::memcpy(d, p->x0 + off, sizeof(double) * p->block_size[k]);
    off += p->block_size[k];
    info->keep_block_data.push_back(d);
    if (blocks_out && addr_of_id) blocks_out->push_back(addr_of_id->at(p->block_id[k]));
  }
  info->linearized_jacobians.resize(p->n, p->n);
  info->linearized_residuals.resize(p->n);
  for (int i = 0; i < p->n; ++i) { info->linearized_residuals(i) = p->r0[i]; for (int j = 0; j < p->n; ++j) info->linearized_jacobians(i, j) = p->J0[p->n * i + j]; }
  return info;
}
void ref_eval_prior(const orc_prior *prior, const double *const *parameters, double *residuals, double **jacobians) {
  MarginalizationInfo *info = info_from_record(prior, nullptr, nullptr);
  MarginalizationFactor f(info);
  f.Evaluate(parameters, residuals, jacobians);
  // (info is leaked on purpose: ~MarginalizationInfo deletes array news with scalar delete)
}

// ---- marginalisation: blocks enumerated as Estimator::optimization does (estimator.cpp:1247-1455), then the
// reference's MarginalizationInfo::{addResidualBlockInfo, preMarginalize, marginalize, getParameterBlocks}.
// out: prior in the REFERENCE's (hash-map dependent) block order; A_out/b_out (optional) are not available here.
int ref_marginalize(const orc_config *cfg, const orc_window *w, const orc_state *s, int mode, orc_prior *out) {
  apply_config(cfg);
  const int F = w->n_frames, WS = F - 1, L = w->n_landmarks;
  // the estimator's parameter arrays (estimator.h: para_Pose[WINDOW_SIZE + 1][SIZE_POSE] ...)
  static double para_Pose[11][7], para_SpeedBias[11][9], para_LegBias[11][4], para_Ex_Pose[2][7], para_Td[1][1];
  std::vector<double> featbuf(L > 0 ? L : 1);
  std::memcpy(para_Pose, s->pose, sizeof(double) * 7 * F);
  std::memcpy(para_SpeedBias, s->speed_bias, sizeof(double) * 9 * F);
  std::memcpy(para_LegBias, s->leg_bias, sizeof(double) * 4 * F);
  std::memcpy(para_Ex_Pose, s->ex_pose, sizeof(double) * 14);
  para_Td[0][0] = s->td[0];
  for (int k = 0; k < L; ++k) featbuf[k] = s->inv_depth[k];
  std::unordered_map<int, double *> addr_of_id;
  std::unordered_map<long, int> id_of_addr;
  auto reg = [&](int id, double *a) { addr_of_id[id] = a; id_of_addr[reinterpret_cast<long>(a)] = id; };
  for (int i = 0; i < F; ++i) { reg(ORC_BLK_POSE * 16 + i, para_Pose[i]); reg(ORC_BLK_SB * 16 + i, para_SpeedBias[i]); reg(ORC_BLK_LB * 16 + i, para_LegBias[i]); }
  reg(ORC_BLK_EX * 16 + 0, para_Ex_Pose[0]); reg(ORC_BLK_EX * 16 + 1, para_Ex_Pose[1]); reg(ORC_BLK_TD * 16, para_Td[0]);
  for (int k = 0; k < L; ++k) reg(ORC_BLK_FEAT * 16 + k, &featbuf[k]);

  ceres::LossFunction *loss_function = new ceres::HuberLoss(g_huber);
  MarginalizationInfo *marginalization_info = new MarginalizationInfo();
  std::vector<double *> last_blocks;
  MarginalizationInfo *last_info = nullptr;
  if (w->prior && w->prior->valid) last_info = info_from_record(w->prior, &last_blocks, &addr_of_id);

  std::unordered_map<long, double *> addr_shift;
  if (mode == 0) {
    if (last_info) {
      std::vector<int> drop_set;
      for (int i = 0; i < (int)last_blocks.size(); ++i)
        if (last_blocks[i] == para_Pose[0] || last_blocks[i] == para_SpeedBias[0] || last_blocks[i] == para_LegBias[0]) drop_set.push_back(i);
      marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(new MarginalizationFactor(last_info), NULL, last_blocks, drop_set));
    }
    if (w->use_leg) {
      if (w->preint[0].sum_dt < 10.0)
        marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(
            new IMULegFactor(il_from_record(&w->preint[0])), NULL,
            std::vector<double *>{para_Pose[0], para_SpeedBias[0], para_LegBias[0], para_Pose[1], para_SpeedBias[1], para_LegBias[1]}, std::vector<int>{0, 1, 2}));
    } else {
      if (w->preint_imu[0].sum_dt < 10.0)
        marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(new IMUFactor(imu_from_record(&w->preint_imu[0])), NULL,
            std::vector<double *>{para_Pose[0], para_SpeedBias[0], para_Pose[1], para_SpeedBias[1]}, std::vector<int>{0, 1}));
    }
    for (int k = 0; k < L; ++k) {
      if (w->lm_start_frame[k] != 0) continue;
      const int o0 = w->lm_obs_offset[k], o1 = w->lm_obs_offset[k + 1];
      const double *f0 = w->obs + 11 * o0;
      for (int o = o0; o < o1; ++o) {
        const int imu_j = o - o0;
        const double *fj = w->obs + 11 * o;
        if (imu_j != 0)
          marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(
              new ProjectionTwoFrameOneCamFactor(v3(f0), v3(fj), Eigen::Vector2d(f0[6], f0[7]), Eigen::Vector2d(fj[6], fj[7]), f0[10], fj[10]), loss_function,
              std::vector<double *>{para_Pose[0], para_Pose[imu_j], para_Ex_Pose[0], &featbuf[k], para_Td[0]}, std::vector<int>{0, 3}));
        if (w->obs_is_stereo[o]) {
          if (imu_j != 0)
            marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(
                new ProjectionTwoFrameTwoCamFactor(v3(f0), v3(fj + 3), Eigen::Vector2d(f0[6], f0[7]), Eigen::Vector2d(fj[8], fj[9]), f0[10], fj[10]), loss_function,
                std::vector<double *>{para_Pose[0], para_Pose[imu_j], para_Ex_Pose[0], para_Ex_Pose[1], &featbuf[k], para_Td[0]}, std::vector<int>{0, 4}));
          else
            marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(
                new ProjectionOneFrameTwoCamFactor(v3(f0), v3(fj + 3), Eigen::Vector2d(f0[6], f0[7]), Eigen::Vector2d(fj[8], fj[9]), f0[10], fj[10]), loss_function,
                std::vector<double *>{para_Ex_Pose[0], para_Ex_Pose[1], &featbuf[k], para_Td[0]}, std::vector<int>{2}));
        }
      }
    }
    for (int i = 1; i <= WS; ++i) {
      addr_shift[reinterpret_cast<long>(para_Pose[i])] = para_Pose[i - 1];
      addr_shift[reinterpret_cast<long>(para_SpeedBias[i])] = para_SpeedBias[i - 1];
      addr_shift[reinterpret_cast<long>(para_LegBias[i])] = para_LegBias[i - 1];
    }
  } else {
    bool has = false;
    for (double *b : last_blocks) if (b == para_Pose[WS - 1]) has = true;
    if (!last_info || !has) return 1;
    std::vector<int> drop_set;
    for (int i = 0; i < (int)last_blocks.size(); ++i) if (last_blocks[i] == para_Pose[WS - 1]) drop_set.push_back(i);
    marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(new MarginalizationFactor(last_info), NULL, last_blocks, drop_set));
    for (int i = 0; i <= WS; ++i) {
      if (i == WS - 1) continue;
      const int t = (i == WS) ? i - 1 : i;
      addr_shift[reinterpret_cast<long>(para_Pose[i])] = para_Pose[t];
      addr_shift[reinterpret_cast<long>(para_SpeedBias[i])] = para_SpeedBias[t];
      if (w->use_leg) addr_shift[reinterpret_cast<long>(para_LegBias[i])] = para_LegBias[t];
    }
  }
  for (int i = 0; i < 2; ++i) addr_shift[reinterpret_cast<long>(para_Ex_Pose[i])] = para_Ex_Pose[i];
  addr_shift[reinterpret_cast<long>(para_Td[0])] = para_Td[0];

  marginalization_info->preMarginalize();
  marginalization_info->marginalize();
  std::vector<double *> parameter_blocks = marginalization_info->getParameterBlocks(addr_shift);

  out->n = marginalization_info->n;
  out->n_blocks = (int)parameter_blocks.size();
  out->valid = marginalization_info->valid ? 1 : 0;
  int off = 0;
  for (int k = 0; k < out->n_blocks; ++k) {
    out->block_id[k] = id_of_addr.at(reinterpret_cast<long>(parameter_blocks[k]));
    out->block_size[k] = marginalization_info->keep_block_size[k];
    out->block_idx[k] = marginalization_info->keep_block_idx[k] - marginalization_info->m;
    std::memcpy(out->x0 + off, marginalization_info->keep_block_data[k], sizeof(double) * out->block_size[k]);
    off += out->block_size[k];
  }
  const int n = out->n;
  for (int i = 0; i < n; ++i) {
    out->r0[i] = marginalization_info->linearized_residuals(i);
    for (int j = 0; j < n; ++j) out->J0[n * i + j] = marginalization_info->linearized_jacobians(i, j);
  }
  return 0;
}

// ---- Utility::R2ypr / ypr2R (utils/utility.h:83-125): the reference's own bodies (header-only, compiled here)
void ref_R2ypr(const double R[9], double ypr[3]) {
  Eigen::Matrix3d M;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M(i, j) = R[3 * i + j];
  Eigen::Vector3d v = Utility::R2ypr(M);
  for (int i = 0; i < 3; ++i) ypr[i] = v(i);
}
void ref_ypr2R(const double ypr[3], double R[9]) {
  Eigen::Matrix3d M = Utility::ypr2R(Eigen::Vector3d(ypr[0], ypr[1], ypr[2]));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = M(i, j);
}
// ---- the gauge fix of Estimator::double2vector (estimator.cpp:905-957; that file needs ROS and OpenCV and cannot be compiled here):
// its statements written down around the reference's Utility::R2ypr / ypr2R, followed by vector2double's re-pack (:848-873). `before` =
// Rs[0] / Ps[0] as they stood when vector2double ran, `after` = the solver's para_* output, overwritten with the fixed states.
void ref_gauge_fix(const orc_state *before, orc_state *after, int F) {
  auto quat = [](const double *p) { return Eigen::Quaterniond(p[6], p[3], p[4], p[5]); };
  Eigen::Matrix3d Rs0 = quat(before->pose).toRotationMatrix();
  Eigen::Vector3d origin_R0 = Utility::R2ypr(Rs0);
  Eigen::Vector3d origin_P0(before->pose[0], before->pose[1], before->pose[2]);
  Eigen::Matrix3d R00 = quat(after->pose).toRotationMatrix();
  Eigen::Vector3d origin_R00 = Utility::R2ypr(R00);
  double y_diff = origin_R0.x() - origin_R00.x();
  Eigen::Matrix3d rot_diff = Utility::ypr2R(Eigen::Vector3d(y_diff, 0, 0));
  if (std::abs(std::abs(origin_R0.y()) - 90) < 1.0 || std::abs(std::abs(origin_R00.y()) - 90) < 1.0) rot_diff = Rs0 * R00.transpose();
  const Eigen::Vector3d P0(after->pose[0], after->pose[1], after->pose[2]);
  for (int i = 0; i < F; ++i) {
    double *pp = after->pose + 7 * i, *sb = after->speed_bias + 9 * i;
    Eigen::Matrix3d Ri = rot_diff * quat(pp).normalized().toRotationMatrix();
    Eigen::Vector3d Pi = rot_diff * Eigen::Vector3d(pp[0] - P0(0), pp[1] - P0(1), pp[2] - P0(2)) + origin_P0;
    Eigen::Vector3d Vi = rot_diff * Eigen::Vector3d(sb[0], sb[1], sb[2]);
    Eigen::Quaterniond q{Ri};
    pp[0] = Pi(0); pp[1] = Pi(1); pp[2] = Pi(2);
    pp[3] = q.x(); pp[4] = q.y(); pp[5] = q.z(); pp[6] = q.w();
    sb[0] = Vi(0); sb[1] = Vi(1); sb[2] = Vi(2);
  }
  for (int c = 0; c < 2; ++c) {
    double *pp = after->ex_pose + 7 * c;
    Eigen::Quaterniond q{quat(pp).normalized().toRotationMatrix()};
    pp[3] = q.x(); pp[4] = q.y(); pp[5] = q.z(); pp[6] = q.w();
  }
}

// ---- what the reference pays per trust-region iteration for factor evaluation alone: every cost function of the window as
// Estimator::optimization adds them (estimator.cpp:1110-1216: the prior, WINDOW_SIZE IMULegFactors, per observation of a landmark
// TwoFrameOneCam / TwoFrameTwoCam / OneFrameTwoCam), each Evaluate()d with residuals AND Jacobians `reps` times, timed here (no ctypes
// call per factor). Returns seconds per pass over the window's factors; n_factors_out = residual blocks evaluated per pass. Ceres'
// own work per iteration (loss correction, local parameterisation, Schur elimination, dogleg) is NOT in it, and Eigen is the build's
// stand-in (shim/Eigen/mini_eigen.h, eager loops): a floor-ish figure for the reference's Evaluate cost, not a Ceres timing.
double ref_time_evaluate(const orc_config *cfg, const orc_window *w, const orc_state *s, int reps, int *n_factors_out) {
  apply_config(cfg);
  const int F = w->n_frames, L = w->n_landmarks;
  static double para_Pose[11][7], para_SpeedBias[11][9], para_LegBias[11][4], para_Ex_Pose[2][7], para_Td[1][1];
  std::vector<double> featbuf(L > 0 ? L : 1);
  std::memcpy(para_Pose, s->pose, sizeof(double) * 7 * F);
  std::memcpy(para_SpeedBias, s->speed_bias, sizeof(double) * 9 * F);
  std::memcpy(para_LegBias, s->leg_bias, sizeof(double) * 4 * F);
  std::memcpy(para_Ex_Pose, s->ex_pose, sizeof(double) * 14);
  para_Td[0][0] = s->td[0];
  for (int k = 0; k < L; ++k) featbuf[k] = s->inv_depth[k];
  struct Blk { ceres::CostFunction *f; std::vector<double *> p; };
  std::vector<Blk> blocks;
  if (w->prior && w->prior->valid) {
    std::unordered_map<int, double *> addr_of_id;
    for (int i = 0; i < F; ++i) { addr_of_id[ORC_BLK_POSE * 16 + i] = para_Pose[i]; addr_of_id[ORC_BLK_SB * 16 + i] = para_SpeedBias[i]; addr_of_id[ORC_BLK_LB * 16 + i] = para_LegBias[i]; }
    addr_of_id[ORC_BLK_EX * 16 + 0] = para_Ex_Pose[0]; addr_of_id[ORC_BLK_EX * 16 + 1] = para_Ex_Pose[1]; addr_of_id[ORC_BLK_TD * 16] = para_Td[0];
    std::vector<double *> last_blocks;
    MarginalizationInfo *info = info_from_record(w->prior, &last_blocks, &addr_of_id);
    blocks.push_back({new MarginalizationFactor(info), last_blocks});
  }
  for (int i = 0; i + 1 < F; ++i) {
    if (w->use_leg)
      blocks.push_back({new IMULegFactor(il_from_record(&w->preint[i])),
                        {para_Pose[i], para_SpeedBias[i], para_LegBias[i], para_Pose[i + 1], para_SpeedBias[i + 1], para_LegBias[i + 1]}});
    else
      blocks.push_back({new IMUFactor(imu_from_record(&w->preint_imu[i])), {para_Pose[i], para_SpeedBias[i], para_Pose[i + 1], para_SpeedBias[i + 1]}});
  }
  for (int k = 0; k < L; ++k) {
    const int o0 = w->lm_obs_offset[k], o1 = w->lm_obs_offset[k + 1], imu_i = w->lm_start_frame[k];
    const double *f0 = w->obs + 11 * o0;
    for (int o = o0; o < o1; ++o) {
      const int imu_j = imu_i + (o - o0);
      const double *fj = w->obs + 11 * o;
      if (imu_j != imu_i)
        blocks.push_back({new ProjectionTwoFrameOneCamFactor(v3(f0), v3(fj), Eigen::Vector2d(f0[6], f0[7]), Eigen::Vector2d(fj[6], fj[7]), f0[10], fj[10]),
                          {para_Pose[imu_i], para_Pose[imu_j], para_Ex_Pose[0], &featbuf[k], para_Td[0]}});
      if (w->obs_is_stereo[o]) {
        if (imu_j != imu_i)
          blocks.push_back({new ProjectionTwoFrameTwoCamFactor(v3(f0), v3(fj + 3), Eigen::Vector2d(f0[6], f0[7]), Eigen::Vector2d(fj[8], fj[9]), f0[10], fj[10]),
                            {para_Pose[imu_i], para_Pose[imu_j], para_Ex_Pose[0], para_Ex_Pose[1], &featbuf[k], para_Td[0]}});
        else
          blocks.push_back({new ProjectionOneFrameTwoCamFactor(v3(f0), v3(fj + 3), Eigen::Vector2d(f0[6], f0[7]), Eigen::Vector2d(fj[8], fj[9]), f0[10], fj[10]),
                            {para_Ex_Pose[0], para_Ex_Pose[1], &featbuf[k], para_Td[0]}});
      }
    }
  }
  *n_factors_out = (int)blocks.size();
  std::vector<double> res(128), jac(128 * 128);
  std::vector<double *> jp(16);
  double sink = 0.0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int r = 0; r < reps; ++r)
    for (Blk &b : blocks) {
      const int nr = b.f->num_residuals();
      int off = 0;
      for (size_t q = 0; q < b.p.size(); ++q) { jp[q] = jac.data() + off; off += nr * b.f->parameter_block_sizes()[q]; }
      b.f->Evaluate(b.p.data(), res.data(), jp.data());
      sink += res[0];
    }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (sink == 12345.678) *n_factors_out = -1;   // (keeps the loop)
  return ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / reps;
}

}  // extern "C"
