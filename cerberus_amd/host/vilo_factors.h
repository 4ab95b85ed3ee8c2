// Host-side C++ mirror of the Cerberus factor interface for the optimisation hot path, forwarding to the C-ABI of
// libvilo_gpu.so (include/vilo_gpu.h). Same names, argument meaning and Evaluate() signature as the reference's
// ceres::CostFunction subclasses, so a maintainer can swap the types in Estimator::optimization():
//
//   reference class (file)                                               this header
//   ceres::CostFunction / SizedCostFunction (ceres/cost_function.h)      vilo::CostFunction / vilo::SizedCostFunction
//   IMULegFactor      src/factor/imu_leg_factor.h:15-26                  vilo::IMULegFactor
//   IMUFactor         src/factor/imu_factor.h:21-28                      vilo::IMUFactor
//   ProjectionTwoFrameOneCamFactor  projectionTwoFrameOneCamFactor.h:21  vilo::ProjectionTwoFrameOneCamFactor
//   ProjectionTwoFrameTwoCamFactor  projectionTwoFrameTwoCamFactor.h     vilo::ProjectionTwoFrameTwoCamFactor
//   ProjectionOneFrameTwoCamFactor  projectionOneFrameTwoCamFactor.h     vilo::ProjectionOneFrameTwoCamFactor
//   MarginalizationFactor           marginalization_factor.h:84-92       vilo::MarginalizationFactor
//   PoseLocalParameterization       pose_local_parameterization.h:14-22  vilo::PoseLocalParameterization
//   ceres::HuberLoss                                                     vilo::HuberLoss
//
// Evaluate() returns true on success (the reference always returns true); on a device error it returns false and
// the message is available from vilo_last_error(). One-factor calls exist for API compatibility; the performance
// path is vilo_solve_windows (WindowSolver below), which never materialises Jacobians.
#pragma once
#include <vector>

#include "../../include/vilo_gpu.h"

namespace vilo {

class CostFunction {
 public:
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
  const std::vector<int> &parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int> *mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int> parameter_block_sizes_;
  int num_residuals_ = 0;
};

template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() {
    set_num_residuals(kNumResiduals);
    *mutable_parameter_block_sizes() = std::vector<int>{Ns...};
  }
};

// ceres::LossFunction::Evaluate(double s, double rho[3])
class HuberLoss {
 public:
  explicit HuberLoss(double a) : a_(a) {}
  void Evaluate(double s, double rho[3]) const { vilo_huber(a_, s, rho); }

 private:
  double a_;
};

class PoseLocalParameterization {
 public:
  explicit PoseLocalParameterization(vilo_ctx *ctx) : ctx_(ctx) {}
  bool Plus(const double *x, const double *delta, double *x_plus_delta) const { return vilo_pose_plus(ctx_, 1, x, delta, x_plus_delta) == 0; }
  bool ComputeJacobian(const double *, double *jacobian) const {   // 7x6 row-major [I6; 0]
    for (int i = 0; i < 42; ++i) jacobian[i] = 0.0;
    for (int i = 0; i < 6; ++i) jacobian[i * 6 + i] = 1.0;
    return true;
  }
  int GlobalSize() const { return 7; }
  int LocalSize() const { return 6; }

 private:
  vilo_ctx *ctx_;
};

// para_Pose[i], para_SpeedBias[i], para_LegBias[i], para_Pose[j], para_SpeedBias[j], para_LegBias[j]
class IMULegFactor : public SizedCostFunction<31, 7, 9, 4, 7, 9, 4> {
 public:
  IMULegFactor(vilo_ctx *ctx, const vilo_preint *il_pre_integration) : ctx_(ctx), pre_(il_pre_integration) {}
  bool Evaluate(double const *const *p, double *r, double **J) const override {
    return vilo_eval_imu_leg(ctx_, 1, pre_, p[0], p[1], p[2], p[3], p[4], p[5], r, J ? J[0] : nullptr, J ? J[1] : nullptr,
                             J ? J[2] : nullptr, J ? J[3] : nullptr, J ? J[4] : nullptr, J ? J[5] : nullptr) == 0;
  }

 private:
  vilo_ctx *ctx_;
  const vilo_preint *pre_;
};

class IMUFactor : public SizedCostFunction<15, 7, 9, 7, 9> {
 public:
  IMUFactor(vilo_ctx *ctx, const vilo_preint_imu *pre_integration) : ctx_(ctx), pre_(pre_integration) {}
  bool Evaluate(double const *const *p, double *r, double **J) const override {
    return vilo_eval_imu(ctx_, 1, pre_, p[0], p[1], p[2], p[3], r, J ? J[0] : nullptr, J ? J[1] : nullptr, J ? J[2] : nullptr,
                         J ? J[3] : nullptr) == 0;
  }

 private:
  vilo_ctx *ctx_;
  const vilo_preint_imu *pre_;
};

struct ProjectionObs {   // constructor arguments of the reference's projection factors
  double obs[12];        // pts_i(3) pts_j(3) velocity_i(2) velocity_j(2) td_i td_j
  ProjectionObs(const double pts_i[3], const double pts_j[3], const double vel_i[2], const double vel_j[2], double td_i, double td_j) {
    for (int k = 0; k < 3; ++k) { obs[k] = pts_i[k]; obs[3 + k] = pts_j[k]; }
    obs[6] = vel_i[0]; obs[7] = vel_i[1]; obs[8] = vel_j[0]; obs[9] = vel_j[1]; obs[10] = td_i; obs[11] = td_j;
  }
};

class ProjectionTwoFrameOneCamFactor : public SizedCostFunction<2, 7, 7, 7, 1, 1> {
 public:
  ProjectionTwoFrameOneCamFactor(vilo_ctx *ctx, const double pts_i[3], const double pts_j[3], const double vel_i[2], const double vel_j[2],
                                 double td_i, double td_j) : ctx_(ctx), o_(pts_i, pts_j, vel_i, vel_j, td_i, td_j) {}
  bool Evaluate(double const *const *p, double *r, double **J) const override {
    return vilo_eval_proj2f1c(ctx_, 1, o_.obs, p[0], p[1], p[2], p[3], p[4], r, J ? J[0] : nullptr, J ? J[1] : nullptr, J ? J[2] : nullptr,
                              J ? J[3] : nullptr, J ? J[4] : nullptr) == 0;
  }

 private:
  vilo_ctx *ctx_;
  ProjectionObs o_;
};

class ProjectionTwoFrameTwoCamFactor : public SizedCostFunction<2, 7, 7, 7, 7, 1, 1> {
 public:
  ProjectionTwoFrameTwoCamFactor(vilo_ctx *ctx, const double pts_i[3], const double pts_j[3], const double vel_i[2], const double vel_j[2],
                                 double td_i, double td_j) : ctx_(ctx), o_(pts_i, pts_j, vel_i, vel_j, td_i, td_j) {}
  bool Evaluate(double const *const *p, double *r, double **J) const override {
    return vilo_eval_proj2f2c(ctx_, 1, o_.obs, p[0], p[1], p[2], p[3], p[4], p[5], r, J ? J[0] : nullptr, J ? J[1] : nullptr,
                              J ? J[2] : nullptr, J ? J[3] : nullptr, J ? J[4] : nullptr, J ? J[5] : nullptr) == 0;
  }

 private:
  vilo_ctx *ctx_;
  ProjectionObs o_;
};

class ProjectionOneFrameTwoCamFactor : public SizedCostFunction<2, 7, 7, 1, 1> {
 public:
  ProjectionOneFrameTwoCamFactor(vilo_ctx *ctx, const double pts_i[3], const double pts_j[3], const double vel_i[2], const double vel_j[2],
                                 double td_i, double td_j) : ctx_(ctx), o_(pts_i, pts_j, vel_i, vel_j, td_i, td_j) {}
  bool Evaluate(double const *const *p, double *r, double **J) const override {
    return vilo_eval_proj1f2c(ctx_, 1, o_.obs, p[0], p[1], p[2], p[3], r, J ? J[0] : nullptr, J ? J[1] : nullptr, J ? J[2] : nullptr,
                              J ? J[3] : nullptr) == 0;
  }

 private:
  vilo_ctx *ctx_;
  ProjectionObs o_;
};

// Dynamic cost function over the kept blocks of a prior (marginalization_factor.cpp:335-395).
class MarginalizationFactor : public CostFunction {
 public:
  MarginalizationFactor(vilo_ctx *ctx, const vilo_prior *marginalization_info) : ctx_(ctx), prior_(marginalization_info) {
    for (int k = 0; k < prior_->n_blocks; ++k) mutable_parameter_block_sizes()->push_back(prior_->block_size[k]);
    set_num_residuals(prior_->n);
  }
  bool Evaluate(double const *const *p, double *r, double **J) const override;

 private:
  vilo_ctx *ctx_;
  const vilo_prior *prior_;
};

// Estimator::optimization() as one call: pack (vector2double) -> solve -> gauge fix (double2vector) -> marginalise.
// marginalization_flag: 0 MARGIN_OLD, 1 MARGIN_SECOND_NEW (estimator.h:64-68).
struct WindowSolver {
  vilo_ctx *ctx;
  vilo_solve_opts opts;
  explicit WindowSolver(vilo_ctx *c) : ctx(c) { vilo_default_solve_opts(&opts); }
  // Returns 0 on success. `window` describes the problem, `state` is updated in place, `next_prior` (optional, with
  // caller-provided x0/J0/r0 buffers) receives last_marginalization_info for the next frame.
  int optimization(const vilo_window_desc &window, vilo_window_state &state, int marginalization_flag, vilo_prior *next_prior,
                   vilo_solve_summary *summary = nullptr);
};

}  // namespace vilo
