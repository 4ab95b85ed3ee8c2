// TEST INFRASTRUCTURE: the part of the Ceres Solver 1.14 public interface that the Cerberus factor classes derive
// from (cost_function.h, sized_cost_function.h, loss_function.h, local_parameterization.h), restated from its
// documentation. No solver here — the trust-region loop is restated in oracle/o_solver.cpp.
#pragma once
#include <cmath>
#include <algorithm>
#include <limits>
#include <vector>
namespace ceres {
typedef int int32;
class CostFunction {
 public:
  CostFunction() : num_residuals_(0) {}
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
  const std::vector<int32> &parameter_block_sizes() const { return parameter_block_sizes_; }
  int num_residuals() const { return num_residuals_; }

 protected:
  std::vector<int32> *mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
  void set_num_residuals(int n) { num_residuals_ = n; }

 private:
  std::vector<int32> parameter_block_sizes_;
  int num_residuals_;
};
template <int kNumResiduals, int... Ns> class SizedCostFunction : public CostFunction {
 public:
  SizedCostFunction() {
    set_num_residuals(kNumResiduals);
    *mutable_parameter_block_sizes() = std::vector<int32>{Ns...};
  }
  virtual ~SizedCostFunction() {}
};
class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
// rho(s) = s for s <= a^2, 2 a sqrt(s) - a^2 otherwise
class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  virtual void Evaluate(double s, double rho[3]) const {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  }

 private:
  const double a_, b_;
};
class CauchyLoss : public LossFunction {
 public:
  explicit CauchyLoss(double a) : b_(a * a), c_(1.0 / b_) {}
  virtual void Evaluate(double s, double rho[3]) const {
    const double sum = 1.0 + s * c_, inv = 1.0 / sum;
    rho[0] = b_ * std::log(sum); rho[1] = std::max(std::numeric_limits<double>::min(), inv); rho[2] = -c_ * (inv * inv);
  }

 private:
  const double b_, c_;
};
class LocalParameterization {
 public:
  virtual ~LocalParameterization() {}
  virtual bool Plus(const double *x, const double *delta, double *x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double *x, double *jacobian) const = 0;
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};
}  // namespace ceres
