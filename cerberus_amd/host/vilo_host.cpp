// Non-inline parts of the host-side mirror (see vilo_factors.h). Links against libvilo_gpu.so only.
#include "vilo_factors.h"

#include <string.h>

#include <vector>

namespace vilo {

bool MarginalizationFactor::Evaluate(double const *const *p, double *r, double **J) const {
  // the C entry point takes the kept blocks concatenated and returns one n x sum(block_size) Jacobian
  int sum_g = 0;
  for (int k = 0; k < prior_->n_blocks; ++k) sum_g += prior_->block_size[k];
  std::vector<double> params(sum_g);
  int off = 0;
  for (int k = 0; k < prior_->n_blocks; ++k) {
    memcpy(&params[off], p[k], sizeof(double) * prior_->block_size[k]);
    off += prior_->block_size[k];
  }
  bool want = false;
  if (J)
    for (int k = 0; k < prior_->n_blocks; ++k) want = want || J[k] != nullptr;
  std::vector<double> Jall(want ? (size_t)prior_->n * sum_g : 0);
  if (vilo_eval_prior(ctx_, 1, prior_, params.data(), r, want ? Jall.data() : nullptr) != 0) return false;
  if (want) {
    off = 0;
    for (int k = 0; k < prior_->n_blocks; ++k) {
      const int gs = prior_->block_size[k];
      if (J[k])
        for (int i = 0; i < prior_->n; ++i) memcpy(J[k] + (size_t)i * gs, &Jall[(size_t)i * sum_g + off], sizeof(double) * gs);
      off += gs;
    }
  }
  return true;
}

int WindowSolver::optimization(const vilo_window_desc &window, vilo_window_state &state, int marginalization_flag, vilo_prior *next_prior,
                               vilo_solve_summary *summary) {
  // solve, double2vector's gauge fix (Rs[0] / Ps[0] of the pre-solve state, estimator.cpp:905-915) and the marginalisation at the
  // result on one device-resident batch; the reference only marginalises full windows (estimator.cpp:1243)
  vilo_solve_summary local;
  const bool marg = next_prior && window.n_frames == VILO_MAX_FRAMES;
  return vilo_optimize_windows(ctx, 1, &window, &state, &opts, marg ? &marginalization_flag : nullptr, marg ? next_prior : nullptr,
                               summary ? summary : &local);
}

}  // namespace vilo
