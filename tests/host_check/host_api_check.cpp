// GPU test program: the C++ mirror classes (cerberus_amd/host/vilo_factors.h) evaluate a factor through the C-ABI and
// WindowSolver::optimization() runs solve + gauge fix + marginalisation on a synthetic window. Prints key numbers that
// tests/test_host_cpp.py compares with the oracle.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../cerberus_amd/host/vilo_factors.h"
#include "../../include/vilo_synth.h"

int main() {
  vilo_config cfg;
  vilo_default_config(&cfg);
  vilo_ctx *ctx = nullptr;
  if (vilo_create(&ctx, &cfg, 0) != 0) { fprintf(stderr, "no GPU\n"); return 2; }
  vilo_synth_params sp;
  vilo_synth_default_params(&sp, 2);
  sp.n_landmarks = 30; sp.seed = 5;
  int32_t n_obs, n_s;
  vilo_synth_sizes(&sp, &n_obs, &n_s);
  const int L = sp.n_landmarks, F = VILO_MAX_FRAMES;
  std::vector<int32_t> lm_start(L), lm_off(L + 1), soff(F);
  std::vector<double> obs(11 * (size_t)n_obs), lin(10 * (F - 1)), pose(7 * F), sb(9 * F), lb(4 * F), ex(14), td(1), lam(L);
  std::vector<double> tpose(7 * F), tsb(9 * F), tlb(4 * F), tlam(L), x0(280), J0(96 * 96), r0(96);
  std::vector<uint8_t> stereo(n_obs);
  std::vector<vilo_sample> samples(n_s);
  vilo_prior prior;
  prior.x0 = x0.data(); prior.J0 = J0.data(); prior.r0 = r0.data();
  vilo_synth_out so = {lm_start.data(), lm_off.data(), obs.data(), stereo.data(), samples.data(), soff.data(), lin.data(),
                       pose.data(), sb.data(), lb.data(), ex.data(), td.data(), lam.data(), tpose.data(), tsb.data(), tlb.data(), tlam.data(), &prior};
  vilo_synth_window(&cfg, &sp, &so);
  std::vector<vilo_preint> pre(F - 1);
  if (vilo_preintegrate(ctx, F - 1, samples.data(), soff.data(), lin.data(), pre.data()) != 0) return 3;
  // one factor through the ceres-shaped adapter
  vilo::IMULegFactor f(ctx, &pre[0]);
  const double *params[6] = {&pose[0], &sb[0], &lb[0], &pose[7], &sb[9], &lb[4]};
  double r[31], J0b[31 * 7];
  double *Js[6] = {J0b, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (!f.Evaluate(params, r, Js)) return 4;
  printf("imu_leg_r0 %.17g %.17g %.17g J00 %.17g\n", r[0], r[1], r[30], J0b[0]);
  vilo_window_desc d;
  d.n_frames = F; d.n_landmarks = L; d.n_obs = n_obs; d.use_leg = 1;
  d.lm_start_frame = lm_start.data(); d.lm_obs_offset = lm_off.data(); d.obs = obs.data(); d.obs_is_stereo = stereo.data();
  d.preint = pre.data(); d.preint_imu = nullptr; d.prior = &prior; d.leg_bias_const = 0; d.ex_const = 0; d.td_const = 1; d.pad = 0;
  vilo_window_state s = {pose.data(), sb.data(), lb.data(), ex.data(), td.data(), lam.data()};
  vilo::WindowSolver solver(ctx);
  solver.opts.fixed_iterations = 1; solver.opts.max_num_iterations = 4;
  std::vector<double> nx0(280), nJ0(96 * 96), nr0(96);
  vilo_prior next;
  next.x0 = nx0.data(); next.J0 = nJ0.data(); next.r0 = nr0.data();
  vilo_solve_summary sum;
  int rc = solver.optimization(d, s, 0, &next, &sum);
  if (rc != 0) { fprintf(stderr, "optimization rc=%d %s\n", rc, vilo_last_error(ctx)); return 5; }
  printf("solve iterations %d successful %d cost %.12g -> %.12g\n", sum.iterations, sum.num_successful, sum.initial_cost, sum.final_cost);
  printf("next_prior n %d blocks %d valid %d first_id %d\n", next.n, next.n_blocks, next.valid, next.block_id[0]);
  printf("pose0 %.12g %.12g %.12g\n", pose[0], pose[1], pose[2]);
  vilo_destroy(ctx);
  return 0;
}
