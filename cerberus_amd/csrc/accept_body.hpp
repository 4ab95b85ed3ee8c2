// Trust-region bookkeeping of one window between two linearisations (Ceres 1.14 TrustRegionMinimizer: IsStepSuccessful,
// Handle{Successful,Unsuccessful,Invalid}Step, DoglegStrategy::Step{Accepted,Rejected,IsInvalid}, the tolerances, max_solver_time): the body
// of k_accept, callable from k_assemble as well (the two are consecutive one-workgroup-per-window launches; fused, an iteration of a small
// batch loses a launch and its latency). Written for 128 threads; the other threads of a larger workgroup pass through its barriers.
#pragma once
#include "solve_common.hpp"
#include "factors.hpp"

using namespace vilo;

struct AcceptParams {
  double min_relative_decrease, function_tolerance, parameter_tolerance;
  int max_num_iterations, fixed_iterations, init_mode, max_solver_time_us;
};

// sums over threads 0 .. 127 (every thread of the workgroup must call): within each of the two waves by shuffles, across them through
// LDS — two workgroup barriers per call whatever the number of sums (the 7-step LDS tree this replaces cost nine per sum, and the
// bookkeeping is a chain of barriers and global round trips: see accept_body). red: 8 doubles.
__device__ __forceinline__ void block_sum128x3(double &a, double &b_, double &c, double *red) {
  const int t = threadIdx.x;
  const bool in = t < 128;
  const double wa = wave_sum(in ? a : 0.0), wb = wave_sum(in ? b_ : 0.0), wc = wave_sum(in ? c : 0.0);
  if (in && (t & 63) == 0) { red[(t >> 6)] = wa; red[2 + (t >> 6)] = wb; red[4 + (t >> 6)] = wc; }
  __syncthreads();
  a = red[0] + red[1]; b_ = red[2] + red[3]; c = red[4] + red[5];
  __syncthreads();
}
__device__ __forceinline__ double block_sum128(double v, double *red) {
  double z0 = 0.0, z1 = 0.0;
  block_sum128x3(v, z0, z1, red);
  return v;
}

// The three decisions of the bookkeeping, each taken by ONE thread of the window's workgroup (shared by accept_body below and by the full
// batch's pose assembly, kernels_asm_full.hip, which runs the same bookkeeping with its loads arranged differently).
// HandleInvalidStep: FAILURE at max_num_consecutive_invalid_steps (5, Ceres default), else DoglegStrategy::StepIsInvalid (mu *= 10, no reuse).
// The candidate pass linearised the unchanged point again (the solver left xc = x): the next step starts from it.
__device__ __forceinline__ void accept_invalid_step(SolverState &st, const AcceptParams &ap) {
  st.num_invalid++;
  if (st.num_invalid >= 5) { st.done = 1; st.termination = 2; }
  else st.mu *= 10.0;
  st.need_lin = 1;
  st.iter++;
  if (st.iter < 64) { st.cost_trace[st.iter] = st.x_cost; st.radius_trace[st.iter] = st.radius; }
  if (st.iter >= ap.max_num_iterations && !st.done) { st.done = 1; st.termination = 0; }
  if (!st.done && ap.max_solver_time_us > 0 && (wall_clock64() - st.t_start) >= 100LL * ap.max_solver_time_us) { st.done = 1; st.termination = 0; }
}
// IterationZero: the initial point's cost
__device__ __forceinline__ void accept_initial_point(SolverState &st, const AcceptParams &ap, double cand, double vis, double imu, double pri) {
  st.x_cost = cand; st.cand_cost = cand; st.vis_cost = vis; st.imu_cost = imu; st.prior_cost = pri;
  st.cost_trace[0] = cand; st.radius_trace[0] = st.radius;
  // a non-finite evaluation at the initial point: ceres::Solve fails in IterationZero ("Residual and Jacobian evaluation
  // failed", ResidualBlock::Evaluate's IsArrayValid) and leaves the parameters alone; this window is done, the others go on
  if (!(cand < 1.7976931348623157e308)) { st.done = 1; st.termination = 2; }
  // "Maximum solver time reached" is checked before every iteration, the first included
  if (!st.done && ap.max_solver_time_us > 0 && (wall_clock64() - st.t_start) >= 100LL * ap.max_solver_time_us) { st.done = 1; st.termination = 0; }
  st.cur ^= 1;   // the pass that gave this cost also linearised the point: its landmark gradients become the current ones
}
// IsStepSuccessful + Handle{Successful,Unsuccessful}Step + DoglegStrategy::Step{Accepted,Rejected}; returns 1 if the candidate is accepted
__device__ __forceinline__ int accept_decide(SolverState &st, const AcceptParams &ap, double cand, double vis, double imu, double pri, double x_cost0, double mcc0) {
  int accepted;
  const double rel = (x_cost0 - cand) / mcc0;
  st.cand_cost = cand;
  st.num_invalid = 0;
  if (rel > ap.min_relative_decrease) {
    accepted = 1;
    st.x_cost = cand; st.vis_cost = vis; st.imu_cost = imu; st.prior_cost = pri;
    if (rel < 0.25) st.radius *= 0.5;
    if (rel > 0.75) st.radius = fmax(st.radius, 3.0 * st.dogleg_step_norm);
    st.mu = fmax(1e-8, 2.0 * st.mu / 10.0);
    st.need_lin = 1;
    st.cur ^= 1;   // the candidate's linearisation (made by the pass that evaluated its cost) becomes the current one
    st.num_successful++;
  } else {
    accepted = 0;
    st.radius *= 0.5;
    st.need_lin = 0;
  }
  st.iter++;
  if (st.iter < 64) { st.cost_trace[st.iter] = st.x_cost; st.radius_trace[st.iter] = st.radius; }
  if (st.iter >= ap.max_num_iterations) { st.done = 1; st.termination = 0; }
  // max_solver_time_in_seconds: "Maximum solver time reached" before the next iteration starts (TrustRegionMinimizer's iteration check)
  if (!st.done && ap.max_solver_time_us > 0 && (wall_clock64() - st.t_start) >= 100LL * ap.max_solver_time_us) { st.done = 1; st.termination = 0; }
  return accepted;
}

// red: 128 doubles (8 used), dxs: VILO_MAX_PRIOR_DIM doubles, accept_sp: one int (LDS); part: null, or (blockDim.x / 96) * 96 doubles of LDS —
// then the prior's H dx is taken by every thread of the workgroup, a slice of the columns each (k_assemble_s: 8 slices of 12 columns; with
// one row per thread the 73 KB of H are 96 dependent loads per thread, 6 round trips of a batch's 16: 15 k of the bookkeeping's 19 k cycles
// at one window). The rows' sums then associate differently: a workgroup size is a property of the kernel, so a window's costs still do not
// depend on the batch it shares.
__device__ __forceinline__ void accept_body(BatchDev &b, const AcceptParams &ap, double *red, double *dxs, int *accept_sp, double *part = nullptr) {
#define accept_s (*accept_sp)

  const int win = blockIdx.x, tid = threadIdx.x;
  const int t128 = tid < 128 ? tid : (1 << 30);   // (the body is written for 128 threads; a larger workgroup's other threads only keep the barriers company)
  SolverState &st = b.st[win];
  if (st.done) return;
  const WinMeta wm = b.win[win];
  // (asking for the state's scalars and the window table ahead of the early exits — the body is a chain of global round trips on a
  // nearly idle chip — bought nothing at one window and cost the full batch's k_accept 10 us: measured, not kept)
  const int step_valid0 = st.step_valid;
  const double x_cost0 = st.x_cost, mcc0 = st.model_cost_change;
  double *x = b.x + (size_t)win * XSTRIDE, *xc = b.xc + (size_t)win * XSTRIDE;
  if (!ap.init_mode && !step_valid0) {
    if (tid == 0) accept_invalid_step(st, ap);
    return;
  }
  // candidate cost = 1/2 (visual rho sums + |imu residuals|^2 + |prior residual|^2). Every load of the three parts is issued before the
  // first barrier (the partial sums wait in registers), the three sums share one exchange.
  double vis = 0.0, imu = 0.0, pri = 0.0, my_hd = 0.0;   // my_hd: row tid of H dx at the candidate (becomes the gradient term when accepted)
  const double c0 = (wm.prior_n > 0) ? b.prior_c0[win] : 0.0;
  for (int c = t128; c < wm.n_waves * VILO_MAX_FRAMES; c += 128) vis += b.chunk_cost[(size_t)wm.wave_off * VILO_MAX_FRAMES + c];
  for (int k = t128; k + 1 < wm.n_frames; k += 128) imu += b.imu_cost[(size_t)win * 10 + k];
  if (wm.prior_n > 0) {
    const int n = wm.prior_n;
    if (tid < wm.prior_nb)
      prior_dx(xc + b.prior_bstate[win * 40 + tid], b.prior_x0 + (size_t)win * 280 + b.prior_bxoff[win * 40 + tid],
               b.prior_bsize[win * 40 + tid], dxs + b.prior_bidx[win * 40 + tid]);
    __syncthreads();
    const double *Hp = b.prior_H + (size_t)win * 96 * 96, *b0 = b.prior_b0 + (size_t)win * 96;
    if (part) {
      const int ns = (int)blockDim.x / 96, sl = tid / 96, row = tid - 96 * sl, per = (n + ns - 1) / ns;
      if (sl < ns) {
        double sacc = 0.0;
        if (row < n) {
          const int q1 = min(n, (sl + 1) * per);
#pragma unroll 4
          for (int q = sl * per; q < q1; ++q) sacc += Hp[(size_t)q * n + row] * dxs[q];
        }
        part[sl * 96 + row] = sacc;
      }
      __syncthreads();
      if (tid < n) {
        double sacc = 0.0;
        for (int s2 = 0; s2 < ns; ++s2) sacc += part[s2 * 96 + tid];
        pri = dxs[tid] * (sacc + 2.0 * b0[tid]);
        my_hd = sacc;
      }
    } else if (tid < n) {   // n <= 96 < 128: one row per thread
      double sacc = 0.0;
#pragma unroll 16
      for (int q = 0; q < n; ++q) sacc += Hp[(size_t)q * n + tid] * dxs[q];
      pri = dxs[tid] * (sacc + 2.0 * b0[tid]);
      my_hd = sacc;
    }
  }
  block_sum128x3(vis, imu, pri, red);
  if (wm.prior_n > 0) pri += c0;
  double cand = 0.5 * (vis + imu + pri);
  if (!isfinite(cand)) cand = 1.7976931348623157e308;
  if (b.rp_on && b.prep_bad) {
    // re-propagation: a covariance integrated at this point that is not positive definite has no sqrt_info — the point cannot be
    // evaluated (its whitened residuals used pivots replaced by 1): treated like a non-finite cost
    for (int k = 0; k + 1 < wm.n_frames; ++k)
      if (!b.imu_skip[(size_t)win * 10 + k] && b.prep_bad[(size_t)win * 10 + k]) cand = 1.7976931348623157e308;
  }
  if (ap.init_mode) {
    if (tid == 0) accept_initial_point(st, ap, cand, vis, imu, pri);
    if (tid < wm.prior_n) b.prior_hd[(size_t)win * 96 + tid] = my_hd;
    return;
  }
  // ambient-space norms for ParameterToleranceReached
  bool converged = false;
  if (!ap.fixed_iterations) {
    double pn = 0.0, ps = 0.0;
    for (int e = t128; e < XSTRIDE; e += 128) {
      bool on = e < XO_TD + 1 && !(e >= XO_EX && e < XO_TD && (wm.const_mask & CONST_EX)) && !(e == XO_TD && (wm.const_mask & CONST_TD)) &&
                !(e >= XO_LB && e < XO_EX && (wm.const_mask & CONST_LB));
      if (on) { pn += x[e] * x[e]; ps += (x[e] - xc[e]) * (x[e] - xc[e]); }
    }
    for (int l = t128; l < wm.L; l += 128) {
      const double a = b.lam[wm.lm_off + l], c = b.lamc[wm.lm_off + l];
      pn += a * a; ps += (a - c) * (a - c);
    }
    double z_ = 0.0;
    block_sum128x3(pn, ps, z_, red);
    const double xn = sqrt(pn), sn = sqrt(ps);
    if (sn <= ap.parameter_tolerance * (xn + ap.parameter_tolerance)) converged = true;
    if (!converged && fabs(x_cost0 - cand) <= ap.function_tolerance * x_cost0) converged = true;
  }
  if (converged) {
    if (tid == 0) { st.done = 1; st.termination = 1; st.cand_cost = cand; }
    return;
  }
  if (tid == 0) accept_s = accept_decide(st, ap, cand, vis, imu, pri, x_cost0, mcc0);
  __syncthreads();
  if (accept_s) {
    if (tid < wm.prior_n) b.prior_hd[(size_t)win * 96 + tid] = my_hd;
    for (int e = t128; e < XSTRIDE; e += 128) x[e] = xc[e];
    for (int l = t128; l < wm.L; l += 128) b.lam[wm.lm_off + l] = b.lamc[wm.lm_off + l];
  }
#undef accept_s
}
