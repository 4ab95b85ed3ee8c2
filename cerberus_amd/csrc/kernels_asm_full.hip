// Assembly of a FULL batch's camera-side normal equations (what Ceres' evaluator hands to its SchurEliminator behind
// estimator.cpp:1221-1236), cut in two kernels by LDS footprint the way the solver is cut by register footprint (kernels_split.hip):
//
//   k_assemble_pose   one 256-thread workgroup per window, 38 KB of LDS and <= 128 registers: FOUR workgroups per CU (the single-kernel assembly of
//                     rounds 2 - 4, k_assemble_c, needed 49 KB and 168 registers: three). The trust-region bookkeeping of the previous step (accept_body.hpp's decisions) runs
//                     first, with the prior's H dx taken from the pre-assembled image this kernel loads into LDS anyway — H itself (59 KB
//                     per window) is never read and the iteration has no k_accept launch. Then the 80 x 80 pose / extrinsic system:
//                     prior image + compact Gram slots (pass by pass of <= 6 frames through a six-slot stage, assemble_compact.hpp's
//                     straight-line pass bodies, scatter by ds_add_f64) + the pose blocks of the IMU factor Grams, written out in
//                     FP64-MFMA accumulator order; gradient, Jacobi scaling, dogleg diagonal and v = D^-2 g of the 80 pose-part dimensions.
//   k_assemble_bias   one 256-thread workgroup per window, 19 KB of LDS: the speed / leg-bias part — A_kk, A_{k+1,k}^T, the coupling rows with
//                     poses k - 1 .. k + 1 frame by frame through a ring of two IMU factor Grams, the prior's rows, gradient / scaling / v
//                     of the 143 dimensions, q = v^T H v of every block that has a speed / leg-bias row, and the sums of both kernels.
//
// Together they write what k_assemble / k_assemble_s write (Cimg, Bimg, cam_gin, cam_scale), so every solver form reads them unchanged; the sums that
// cross the two kernels (q, |D^-1 g|^2, max |g|) are added pose part first.
#include "solve_common.hpp"
#include "assemble_compact.hpp"
#include "accept_body.hpp"
#include "wave_common.hpp"
#include "lin_common.hpp"

using namespace vilo;

#define AF_THREADS 256
#define AF_STAGE (AC_PASS * VILO_GRAMC)   // 1104 doubles: the slots of a pass; behind them a slot of zeros (assemble_compact.hpp AC_ZSLOT)
#define BI_SCAL_POSE (BI_SCAL + 3)         // k_assemble_pose's share of the three sums (k_assemble_bias adds its own and writes BI_SCAL + 0 .. 2)

// scratch of the bookkeeping inside the stage (doubles), before the first pass of slots arrives
#define AS_RED 0      // [12]  three sums x four waves
#define AS_FLAG 12    // [2]   ints: proceed, accepted
#define AS_PB 16      // [2][16] BP^T dx partial sums of the two waves that hold pose rows
#define AS_DXS 48     // [96]  dx by prior dimension
#define AS_DXC 144    // [96]  dx by camera dimension (0 .. 79 pose part, 80 .. 92 the prior's speed / leg-bias frame)
#define AS_B0C 240    // [96]  b0 by camera dimension
#define AS_HDC 336    // [96]  H dx by camera dimension
#define AS_XC 432     // [240] the candidate state
#define AS_X0 672     // [280] the prior's linearisation point; afterwards
#define AS_PART 672   // [2][80] halves of the pose rows' sums
#define AS_END 952

__global__ void __launch_bounds__(AF_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_assemble_pose(BatchDev b, int jacobi_scaling, double min_lm_diagonal, double max_lm_diagonal, AcceptParams ap) {
  __shared__ double Cl[CL_N];
  __shared__ double stage[AF_STAGE + VILO_GRAMC];
  __shared__ double Rt[12 * 9];
  __shared__ double gl[80], vS[80];
  __shared__ unsigned pass_tab[2 * 64];   // (a chunk has at most 11 frames: two passes)
  __shared__ int pass_cnt;
  __shared__ short inv_pmap[CD_N];
  __shared__ short pml[96];
  __shared__ unsigned char act[80];
  const int win = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  SolverState &st = b.st[win];
  const double *pd = b.prior_dense + (size_t)win * PD_N;
  const double *xc = b.xc + (size_t)win * XSTRIDE;
  double *bimg = b.Bimg + (size_t)win * BI_N;
  int *flags = (int *)(stage + AS_FLAG);
  PCLK(if (tid == 0) st.phase_clk[36] = clock64());

  // ---- Everything this phase reads from global memory is asked for at once, with addresses that depend on the window's index only: the
  //      bookkeeping is otherwise a chain of dependent round trips (state -> window table -> prior block tables -> the blocks' states), 3 - 5 k
  //      cycles each on a busy chip. The candidate state and the prior's linearisation point go to LDS whole; dx of the prior's blocks is
  //      formed from there once the block tables have arrived. (Rows of the per-window tables exist for every window, prior or not.) ----
  double pv[13];
#pragma unroll
  for (int u = 0; u < 13; ++u) {
    const int e = min(tid + AF_THREADS * u, CL_N - 1);
    int row = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    while (((row + 1) * (row + 2)) / 2 <= e) ++row;
    while ((row * (row + 1)) / 2 > e) --row;
    pv[u] = pd[PD_C + row * PD_CLD + (e - (row * (row + 1)) / 2)];
  }
  // prior: its coupling rows by pose column (thread c < 80: column c of the 13 rows)
  double bpr[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) bpr[i] = (tid < 80) ? pd[PD_BP + i * 80 + tid] : 0.0;
  const double my_xc = (tid < XSTRIDE) ? xc[tid] : 0.0;   // (the candidate: copied to x if it is accepted)
  const double my_x0a = b.prior_x0[(size_t)win * 280 + tid], my_x0b = (tid < 280 - AF_THREADS) ? b.prior_x0[(size_t)win * 280 + AF_THREADS + tid] : 0.0;
  const double my_b0 = (tid < 96) ? b.prior_b0[(size_t)win * 96 + tid] : 0.0;
  const int my_cd = (tid < 96) ? b.prior_map[(size_t)win * 96 + tid] : 0;
  int bl_state = 0, bl_xoff = 0, bl_size = 0, bl_idx = 0;
  if (tid < 40) {
    bl_state = b.prior_bstate[win * 40 + tid]; bl_xoff = b.prior_bxoff[win * 40 + tid];
    bl_size = b.prior_bsize[win * 40 + tid]; bl_idx = b.prior_bidx[win * 40 + tid];
  }
  double imu = (tid < 10) ? b.imu_cost[(size_t)win * 10 + tid] : 0.0;
  if (st.done) return;
  const WinMeta wm = b.win[win];
  const int F = wm.n_frames, cmask = wm.const_mask, kb = wm.pad, pn = wm.prior_n;
  const int step_valid0 = st.step_valid;
  const double x_cost0 = st.x_cost, mcc0 = st.model_cost_change;
  const int mode = ap.init_mode ? 0 : (step_valid0 ? 1 : 2);   // 0 the initial point, 1 a candidate to judge, 2 an invalid step
  const double c0 = (pn > 0) ? b.prior_c0[win] : 0.0;
  // (what needs the window table: the visual cost's partial sums, the chunk table, the prior's speed / leg-bias block — thread 160 + i: row i, 13 entries)
  const bool has_pb = pn > 0 && kb >= 0;
  if (has_pb && tid >= 160 && tid < 173) {
#pragma unroll
    for (int i = 0; i < 13; ++i) bpr[i] = pd[PD_AD + 169 * kb + (tid - 160) * 13 + i];
  }
  double vis = 0.0, pri = 0.0;
  for (int c = tid; c < wm.n_waves * VILO_MAX_FRAMES; c += AF_THREADS) vis += b.chunk_cost[(size_t)wm.wave_off * VILO_MAX_FRAMES + c];
  if (tid + 1 >= F) imu = 0.0;
  if (wv == 0) {
    // pass table (s | t0 << 4 | np << 8 | first slot << 12): lane ch splits its chunk of km frames into ceil(km / AC_PASS) passes of nearly
    // equal length; the passes of all chunks in chunk order (a prefix sum over the wave's lanes)
    const int nch = min(wm.n_chunks, 64);
    int s_c = 0, km = 0, sl0 = 0;
    if (lane < nch) {
      const ChunkMeta cm = b.chunk[wm.chunk_off + lane];
      s_c = cm.s; km = cm.kmax; sl0 = cm.gram_off - wm.gram_off;
    }
    const int npass = (km + AC_PASS - 1) / AC_PASS;
    int incl = npass;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    const int first = incl - npass;
    if (npass > 0) {
      const int base = km / npass, rem = km - base * npass;
      for (int i = 0; i < npass; ++i) {
        const int t0 = i * base + min(i, rem), np_ = base + (i < rem ? 1 : 0);
        pass_tab[first + i] = (unsigned)s_c | ((unsigned)t0 << 4) | ((unsigned)np_ << 8) | ((unsigned)(sl0 + t0) << 12);
      }
    }
    if (lane == 63) pass_cnt = incl;
  }
  if (tid < 96) stage[AS_DXC + tid] = 0.0;
  if (tid < VILO_GRAMC) stage[AF_STAGE + tid] = 0.0;
  if (tid < CD_N) inv_pmap[tid] = -1;
  if (tid < 80) act[tid] = cd_active(tid, F, cmask) ? 1 : 0;
  if (tid < XSTRIDE) stage[AS_XC + tid] = my_xc;
  stage[AS_X0 + tid] = my_x0a;
  if (tid < 280 - AF_THREADS) stage[AS_X0 + AF_THREADS + tid] = my_x0b;
  if (tid < pn) pml[tid] = (short)my_cd;
#pragma unroll
  for (int u = 0; u < 13; ++u) {
    const int e = tid + AF_THREADS * u;
    if (e < CL_N) Cl[e] = pv[u];
  }
  __syncthreads();
  const int pass_n = pass_cnt;
  PCLK(if (tid == 0) st.phase_clk[12] = clock64());
  if (pn > 0 && tid < wm.prior_nb) {
    // dx of this thread's block of the prior at the candidate, by prior dimension and by camera dimension
    double *dxs = stage + AS_DXS + bl_idx;
    prior_dx(stage + AS_XC + bl_state, stage + AS_X0 + bl_xoff, bl_size, dxs);
    const int nloc = bl_size == 7 ? 6 : bl_size;
    for (int i = 0; i < nloc; ++i) {
      const int cd = pml[bl_idx + i];
      stage[AS_DXC + (cd < CD_B0 ? cd : 80 + (cd - CD_B0 - 13 * kb))] = dxs[i];
    }
  }
  if (tid >= 64 && tid < 64 + pn) {
    const int p = tid - 64, cd = pml[p];
    inv_pmap[cd] = (short)p;
  }
  if (tid < pn) stage[AS_B0C + (my_cd < CD_B0 ? my_cd : 80 + (my_cd - CD_B0 - 13 * kb))] = my_b0;
  if (tid >= 224 && tid < 235) {
    // rotation matrices of the window's frames at the point the slots were linearised at: every way past the bookkeeping (the initial point,
    // an accepted candidate, an invalid step) leaves x = xc
    const m3 R = qR(ldq_pose(stage + AS_XC + XO_POSE + 7 * (tid - 224)));
#pragma unroll
    for (int q = 0; q < 9; ++q) Rt[9 * (tid - 224) + q] = R.a[q];
  } else if (tid >= 235 && tid < 244) {
    Rt[99 + (tid - 235)] = ((tid - 235) % 4 == 0) ? 1.0 : 0.0;
  }
  __syncthreads();
  // ---- H dx out of the image (symmetric: the packed lower triangle in LDS serves rows and columns) ----
  const double *dxc = stage + AS_DXC;
  if (pn > 0) {
    if (tid < 160) {
      const int c = tid % 80, h = tid / 80;
      double sacc = 0.0;
      const int rowc = cl_pos(c, 0);
#pragma unroll 8
      for (int q = 40 * h; q < 40 * h + 40; ++q) sacc += Cl[q <= c ? rowc + q : cl_pos(q, 0) + c] * dxc[q];   // (q: the same in every lane — its row start is a scalar)
      stage[AS_PART + 80 * h + c] = sacc;
    }
    if (tid < 128) {
      // threads 0 .. 79 hold column c of the prior's coupling rows: their share of the 13 speed / leg-bias rows, summed over the wave
      const double dc = (tid < 80) ? dxc[tid] : 0.0;
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        const double sred = wave_sum_dpp((tid < 80) ? bpr[i] * dc : 0.0);
        if (lane == 0) stage[AS_PB + 16 * wv + i] = sred;
      }
    }
  }
  __syncthreads();
  PCLK(if (tid == 0) st.phase_clk[13] = clock64());
  if (pn > 0) {
    if (tid < 80) {
      double sacc = stage[AS_PART + tid] + stage[AS_PART + 80 + tid];
#pragma unroll
      for (int i = 0; i < 13; ++i) sacc += bpr[i] * dxc[80 + i];
      stage[AS_HDC + tid] = sacc;
    } else if (tid >= 160 && tid < 173) {
      double sacc = stage[AS_PB + (tid - 160)] + stage[AS_PB + 16 + (tid - 160)];
#pragma unroll
      for (int j = 0; j < 13; ++j) sacc += bpr[j] * dxc[80 + j];
      stage[AS_HDC + 80 + (tid - 160)] = sacc;
    }
  }
  __syncthreads();
  double my_hd = 0.0;
  if (tid < pn) {
    my_hd = stage[AS_HDC + (my_cd < CD_B0 ? my_cd : 80 + (my_cd - CD_B0 - 13 * kb))];
    pri = stage[AS_DXS + tid] * (my_hd + 2.0 * my_b0);
  }
  // candidate cost = 1/2 (visual rho sums + |imu residuals|^2 + |prior residual|^2): waves in fixed order
  {
    const double wa = wave_sum_dpp(vis), wb = wave_sum_dpp(imu), wc = wave_sum_dpp(pri);
    if (lane == 0) { stage[AS_RED + wv] = wa; stage[AS_RED + 4 + wv] = wb; stage[AS_RED + 8 + wv] = wc; }
  }
  bool converged_chk = (mode == 1 && !ap.fixed_iterations);
  double pnrm = 0.0, psq = 0.0;
  if (converged_chk) {
    // ambient-space norms for ParameterToleranceReached
    const double *x = b.x + (size_t)win * XSTRIDE;
    for (int e = tid; e < XSTRIDE; e += AF_THREADS) {
      const bool on = e < XO_TD + 1 && !(e >= XO_EX && e < XO_TD && (cmask & CONST_EX)) && !(e == XO_TD && (cmask & CONST_TD)) &&
                      !(e >= XO_LB && e < XO_EX && (cmask & CONST_LB));
      if (on) { pnrm += x[e] * x[e]; psq += (x[e] - xc[e]) * (x[e] - xc[e]); }
    }
    for (int l = tid; l < wm.L; l += AF_THREADS) {
      const double a = b.lam[wm.lm_off + l], c = b.lamc[wm.lm_off + l];
      pnrm += a * a; psq += (a - c) * (a - c);
    }
    pnrm = wave_sum_dpp(pnrm); psq = wave_sum_dpp(psq);
    if (lane == 0) { stage[AS_PB + wv] = pnrm; stage[AS_PB + 4 + wv] = psq; }   // (the coupling rows' partial sums are consumed)
  }
  __syncthreads();
  PCLK(if (tid == 0) st.phase_clk[14] = clock64());
  if (tid == 0) {
    vis = ((stage[AS_RED + 0] + stage[AS_RED + 1]) + stage[AS_RED + 2]) + stage[AS_RED + 3];
    imu = ((stage[AS_RED + 4] + stage[AS_RED + 5]) + stage[AS_RED + 6]) + stage[AS_RED + 7];
    pri = ((stage[AS_RED + 8] + stage[AS_RED + 9]) + stage[AS_RED + 10]) + stage[AS_RED + 11];
    if (pn > 0) pri += c0;
    double cand = 0.5 * (vis + imu + pri);
    if (!isfinite(cand)) cand = 1.7976931348623157e308;
    if (b.rp_on && b.prep_bad) {
      // re-propagation: a covariance integrated at this point that is not positive definite has no sqrt_info — the point cannot be
      // evaluated: treated like a non-finite cost (accept_body)
      for (int k = 0; k + 1 < F; ++k)
        if (!b.imu_skip[(size_t)win * 10 + k] && b.prep_bad[(size_t)win * 10 + k]) cand = 1.7976931348623157e308;
    }
    int accepted = 0;
    if (mode == 2) accept_invalid_step(st, ap);
    else if (mode == 0) accept_initial_point(st, ap, cand, vis, imu, pri);
    else {
      bool converged = false;
      if (converged_chk) {
        const double xn = sqrt(((stage[AS_PB + 0] + stage[AS_PB + 1]) + stage[AS_PB + 2]) + stage[AS_PB + 3]);
        const double sn = sqrt(((stage[AS_PB + 4] + stage[AS_PB + 5]) + stage[AS_PB + 6]) + stage[AS_PB + 7]);
        if (sn <= ap.parameter_tolerance * (xn + ap.parameter_tolerance)) converged = true;
        if (!converged && fabs(x_cost0 - cand) <= ap.function_tolerance * x_cost0) converged = true;
      }
      if (converged) { st.done = 1; st.termination = 1; st.cand_cost = cand; }
      else accepted = accept_decide(st, ap, cand, vis, imu, pri, x_cost0, mcc0);
    }
    flags[0] = (!st.done && st.need_lin) ? 1 : 0;
    flags[1] = accepted;
  }
  __syncthreads();
  PCLK(if (tid == 0) st.phase_clk[15] = clock64());
  const int proceed = flags[0], accepted = flags[1];
  if (accepted) {
    // the candidate becomes the current point
    if (tid < XSTRIDE) b.x[(size_t)win * XSTRIDE + tid] = my_xc;
    for (int l = tid; l < wm.L; l += AF_THREADS) b.lam[wm.lm_off + l] = b.lamc[wm.lm_off + l];
  }
  if ((accepted || mode != 1) && tid < pn) b.prior_hd[(size_t)win * 96 + tid] = my_hd;   // H dx at the point the next gradient is formed at (k_assemble_bias reads it)
  if (!proceed) return;
  // the prior's share (b0 + H dx) of the gradient of its speed / leg-bias dimensions, where k_assemble_bias starts that gradient from
  if (has_pb && tid >= 160 && tid < 173) {
    const int cd = CD_B0 + 13 * kb + (tid - 160);
    b.cam_gin[(size_t)win * CD_N + cd] = (inv_pmap[cd] >= 0) ? stage[AS_B0C + 80 + (tid - 160)] + stage[AS_HDC + 80 + (tid - 160)] : 0.0;
  }
  PCLK(if (tid == 0) st.phase_clk[37] = clock64());

  // gradient of the pose part starts from the prior's b0 + H dx
  if (tid < 80) gl[tid] = (pn > 0 && inv_pmap[tid] >= 0) ? stage[AS_B0C + tid] + stage[AS_HDC + tid] : 0.0;
  const double *gs = b.gram + (size_t)wm.gram_off * VILO_GRAMC;
  auto rmw = [&](int hi, int lo, double v) { lds_add(&Cl[cl_pos(hi, lo)], v); };   // hi >= lo; a target has one owner thread (assemble_compact.hpp)
  auto add_at = [&](int pos, double v) { lds_add(&Cl[pos], v); };
  auto gadd = [&](int cd, double v) { lds_add(&gl[cd], v); };
  // ---- compact Gram slots, pass by pass: the pass's slots (<= 6 x 184 doubles, contiguous) come into the stage with coalesced loads — every
  //      byte once — and the owner threads gather from there. TWO passes' slots are in flight in registers (pfa: even passes, pfb: odd): with
  //      one pass ahead every pass waited for its own load (a body is ~1 k cycles, a round trip under load 3 - 4 k) ----
  {
    const int n_pass = pass_n;
    double pfa[5], pfb[5];
    auto prefetch = [&](int p, double (&pf)[5]) {
      const unsigned pt = pass_tab[min(p, n_pass - 1)];   // (past the end: the last pass again — an unconditional load keeps the count of loads in flight static)
      const int n = (int)((pt >> 8) & 15) * VILO_GRAMC;
      const double *src = gs + (size_t)(pt >> 12) * VILO_GRAMC;
#pragma unroll
      for (int q = 0; q < 5; ++q) pf[q] = src[min(tid + AF_THREADS * q, n - 1)];   // (no predicate: a load under a branch costs the compiler its count of the loads in flight; what lies beyond the pass's slots in the stage is never read)
    };
    if (n_pass > 0) { prefetch(0, pfa); prefetch(1, pfb); }
    __syncthreads();   // (gl, and the bookkeeping's scratch in the stage is dead)
    // Every wave runs the SAME loop over the passes — two workgroup barriers per pass, the stage filled by all 256 threads — but with its own
    // class bodies, as four instantiations of the loop: the bodies' per-lane constants (entry indices, signs, targets) are loop invariants
    // the compiler keeps in registers, and in one loop for all classes every lane would hold the constants of all of them (185 registers
    // spilled); as four paths the kernel needs the largest path's. (The barriers of the four paths are different instructions that pair up
    // by count: every path executes exactly two per pass.)
    auto pass_loop = [&](auto body) {
      long long c_wait = 0, c_fill = 0, c_body = 0;   // (profiling build: where a wave's cycles of the pass loop go)
      auto step = [&](int p, double (&pf)[5]) {
        const unsigned pt = pass_tab[p];
        const long long k0 = pclk64();
        lds_barrier();   // (the previous pass's readers are done)
        const long long k1 = pclk64();
#pragma unroll
        for (int q = 0; q < 5; ++q) { const int e = tid + AF_THREADS * q; if (e < AF_STAGE) stage[e] = pf[q]; }
        lds_barrier();
        const long long k2 = pclk64();
        prefetch(p + 2, pf);
        body((int)(pt & 15), (int)((pt >> 4) & 15), (int)((pt >> 8) & 15));
        PCLK(const long long k3 = clock64(); c_wait += k1 - k0; c_fill += k2 - k1; c_body += k3 - k2);
      };
      for (int p = 0; p < n_pass; p += 2) {
        step(p, pfa);
        if (p + 1 < n_pass) step(p + 1, pfb);
      }
      PCLK(if (lane == 0) { st.phase_clk[47 + 3 * wv] = c_wait; st.phase_clk[48 + 3 * wv] = c_fill; st.phase_clk[49 + 3 * wv] = c_body; });
    };
    if (wv == 0) {
      pass_loop([&](int s_, int t0_, int np_) { ac_pass_w0(lane, s_, t0_, np_, stage, Rt, add_at, gadd); });
    } else if (wv == 1) {
      // the {tic, tic2}^2 entries, three lanes per entry (each a third of the pass's frames), partial sums added in lane order
      const int q = lane % 21, grp = min(lane / 21, 2);
      pass_loop([&](int s_, int t0_, int np_) {
        const double part = (lane < 63) ? ac_t8_pass(q, grp, s_, t0_, np_, stage, Rt) : 0.0;
        const double p1 = __shfl(part, q + 21, 64), p2 = __shfl(part, q + 42, 64);
        if (lane < 21) ac_t8_apply(q, (part + p1) + p2, rmw);
      });
    } else if (wv == 2) {
      // the tic / tic2 x pose entries, two lanes per entry (alternate frames), the sums over the frames added in lane order (lanes 0 .. 35);
      // {theta_ic, theta_ic2, r}^2 on lanes 36 .. 63
      const int q = lane % 18, par = min(lane / 18, 1);
      pass_loop([&](int s_, int t0_, int np_) {
        double s5 = 0.0, s6 = 0.0;
        if (lane < 36) ac_t56_pass(q, par, s_, t0_, np_, stage, Rt, rmw, s5, s6);
        const double o5 = __shfl(s5, q + 18, 64), o6 = __shfl(s6, q + 18, 64);
        if (lane < 18) ac_t56_apply(q, s_, s5 + o5, s6 + o6, rmw);
        ac_pass_t4(lane - 36, s_, t0_, np_, stage, Rt, rmw, gadd);
      });
    } else {
      pass_loop([&](int s_, int t0_, int np_) { ac_pass_t7(lane, s_, t0_, np_, stage, Rt, rmw, gadd); });
    }
  }
  PCLK(if (tid == 0) st.phase_clk[38] = clock64());
  // ---- pose blocks of the IMU factor Grams (factor k: frames k, k + 1; packed 39 x 39 [pose_i 6 | speed / leg-bias_i 13 | pose_j 6 | ... | r]):
  //      I1 pose_i x pose_i (21, twin pose_j x pose_j), I3 the pose gradient (6, twin), I4 pose_i x pose_j (36): 63 owner threads, straight
  //      from global memory, all of a thread's 20 loads in flight ----
  if (tid >= 64 && tid < 127) {
    int pa = 0, pbc = 0, pcls = 0;
    const int q = tid - 64;
    if (q < 21) { pcls = 1; int rem = q; while (rem >= 6 - pa) { rem -= 6 - pa; ++pa; } pbc = pa + rem; }
    else if (q < 27) { pcls = 3; pa = q - 21; pbc = 38; }
    else { pcls = 4; pa = (q - 27) / 6; pbc = 19 + (q - 27) % 6; }
    const int pe1 = tri39(pa, pbc), pe2 = (pcls == 4) ? pe1 : tri39(pa + 19, pcls == 3 ? 38 : pbc + 19);
    const double *igram = b.imu_gram + (size_t)win * 10 * 780;
    double vm[10], vt[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { vm[k] = igram[780 * k + pe1]; vt[k] = igram[780 * k + pe2]; }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      if (k < F - 1) {
        if (pcls == 1) { rmw(6 * k + pbc, 6 * k + pa, vm[k]); rmw(6 * (k + 1) + pbc, 6 * (k + 1) + pa, vt[k]); }
        else if (pcls == 3) { gadd(6 * k + pa, vm[k]); gadd(6 * (k + 1) + pa, vt[k]); }
        else rmw(6 * (k + 1) + (pbc - 19), 6 * k + pa, vm[k]);
      }
    }
  }
  __syncthreads();
  PCLK(if (tid == 0) st.phase_clk[39] = clock64());
  // ---- gradient, Jacobi scaling 1 / (1 + sqrt(H_ii)) frozen at the first linearisation, dogleg diagonal clamp(diag, 1e-6, 1e32) in the
  //      scaled space, v = D^-2 g (Ceres 1.14 TrustRegionMinimizer / DoglegStrategy) of the 80 pose-part dimensions ----
  double part_gn = 0.0, part_gmax = 0.0, part_q = 0.0;
  if (tid < 80) {
    const int cd = tid;
    const bool on = act[cd];
    const double ge = on ? gl[cd] : 0.0, hdv = on ? Cl[cl_pos(cd, cd)] : 1.0;
    b.cam_gin[(size_t)win * CD_N + cd] = ge;
    double d = 1.0, ve = 0.0;
    if (on) {
      double sc;
      double *cs = b.cam_scale + (size_t)win * CD_N + cd;
      if (!st.scale_ready) { sc = jacobi_scaling ? 1.0 / (1.0 + sqrt(hdv)) : 1.0; *cs = sc; }
      else sc = *cs;
      const double d2 = fmin(fmax(sc * sc * hdv, min_lm_diagonal), max_lm_diagonal);
      d = d2 / (sc * sc);
      ve = ge / d;
    }
    vS[cd] = ve;
    bimg[BI_DH2 + cd] = d;
    bimg[BI_V + cd] = ve;
    part_gn = ge * ve;
    part_gmax = fabs(ge);
  }
  __syncthreads();
  PCLK(if (tid == 0) st.phase_clk[41] = clock64());
  // ---- pose system out in accumulator order; q = v^T H v of its rows is summed while the blocks pass through registers ----
#pragma unroll
  for (int t = 0; t < 15; ++t) {
    const int rr = tid >> 4, cc = tid & 15;
    const int row = 16 * c_tI[t] + rr, col = 16 * c_tJ[t] + cc;
    const double v = (act[row] && act[col]) ? Cl[cl_pos(max(row, col), min(row, col))] : (row == col ? 1.0 : 0.0);
    b.Cimg[(size_t)win * CIMG_N + 256 * t + tid] = v;
    part_q += ((c_tI[t] == c_tJ[t]) ? 1.0 : 2.0) * vS[row] * v * vS[col];   // (a diagonal tile holds both triangles)
  }
  PCLK(if (tid == 0) st.phase_clk[42] = clock64());
  part_q = wave_sum_dpp(part_q); part_gn = wave_sum_dpp(part_gn); part_gmax = wave_max_dpp(part_gmax);
  __syncthreads();   // (the stage's last readers are long done; reuse its head for the sums)
  if (lane == 0) { stage[wv] = part_q; stage[4 + wv] = part_gn; stage[8 + wv] = part_gmax; }
  __syncthreads();
  if (tid == 0) {
    bimg[BI_SCAL_POSE + 0] = ((stage[0] + stage[1]) + stage[2]) + stage[3];
    bimg[BI_SCAL_POSE + 1] = ((stage[4] + stage[5]) + stage[6]) + stage[7];
    bimg[BI_SCAL_POSE + 2] = fmax(fmax(stage[8], stage[9]), fmax(stage[10], stage[11]));
  }
  PCLK(if (tid == 0) st.phase_clk[45] = clock64());
}

// =================================================================================================
// k_assemble_bias
// =================================================================================================
__global__ void __launch_bounds__(AF_THREADS) k_assemble_bias(BatchDev b, int jacobi_scaling, double min_lm_diagonal, double max_lm_diagonal) {
  __shared__ double ring[2 * 780];
  __shared__ double vS[CD_N], red[12];
  __shared__ unsigned char act[CD_N];
  const int win = blockIdx.x, tid = threadIdx.x;
  const SolverState &st = b.st[win];
  if (st.done || !st.need_lin) return;
  const WinMeta wm = b.win[win];
  const int F = wm.n_frames, cmask = wm.const_mask, kb = wm.pad, pn = wm.prior_n;
  const double *igram = b.imu_gram + (size_t)win * 10 * 780;
  const double *pd = b.prior_dense + (size_t)win * PD_N;
  double *bimg = b.Bimg + (size_t)win * BI_N;
  PCLK(if (tid == 0) b.st[win].phase_clk[40] = clock64());
  // two factors' Grams in flight in registers (gpa: even frames, gpb: odd): a load has two trips of the frame loop to arrive. No predicate on
  // the loads (entry and factor clamped): a load under a branch costs the compiler its count of the loads in flight, and every trip would
  // wait for everything (all 10 Grams of a window exist in memory, zeros for the intervals it does not have)
  double gpa[4], gpb[4];
  auto prefetch = [&](int k, double (&gp)[4]) {
    const double *src = igram + min(k, 9) * 780;
#pragma unroll
    for (int u = 0; u < 4; ++u) gp[u] = src[min(tid + AF_THREADS * u, 779)];
  };
  prefetch(0, gpa);
  prefetch(1, gpb);
  // ---- diagonal, gradient, Jacobi scaling 1 / (1 + sqrt(H_ii)) (frozen at the first linearisation), dogleg diagonal and v = D^-2 g of the 143
  //      speed / leg-bias dimensions UP FRONT, thread d = dimension: the four Gram entries a dimension needs (its diagonal entry and its
  //      gradient entry in the factor before and in the factor after its frame) are gathered straight from global memory under the first
  //      Grams' loads, so that q = v^T H v is summed while the blocks pass through the frame loop — nothing is read back ----
  double part_gn = 0.0, part_gmax = 0.0, part_q = 0.0;
  if (tid < CD_N - CD_B0) {
    const int cd = CD_B0 + tid, k = min(tid / 13, 10), i = tid - 13 * (tid / 13);
    const bool on = cd_active(cd, F, cmask);
    const bool has_i = k < F - 1, has_j = k >= 1;
    const double *Gi = igram + 780 * min(k, 9), *Gj = igram + 780 * max(k - 1, 0);
    const double hi_ = Gi[tri39(6 + i, 6 + i)], gi_ = Gi[tri39(6 + i, 38)], hj_ = Gj[tri39(25 + i, 25 + i)], gj_ = Gj[tri39(25 + i, 38)];
    const bool pr = on && k == kb;   // (kb < 0: no prior on any speed / leg-bias block)
    const double ph = pr ? pd[PD_AD + 169 * max(kb, 0) + 14 * i] : 0.0;
    const double pg = (pr && pn > 0) ? b.cam_gin[(size_t)win * CD_N + cd] : 0.0;   // the prior's b0 + H dx of this dimension (k_assemble_pose left it there)
    double *cs = b.cam_scale + (size_t)win * CD_N + cd;
    const double sc_old = (on && st.scale_ready) ? *cs : 1.0;
    double h = 1.0, g = 0.0;
    if (on) {
      h = ph; g = pg;
      if (has_i) { h += hi_; g += gi_; }
      if (has_j) { h += hj_; g += gj_; }
    }
    b.cam_gin[(size_t)win * CD_N + cd] = g;
    double d = 1.0, ve = 0.0;
    if (on) {
      double sc = sc_old;
      if (!st.scale_ready) { sc = jacobi_scaling ? 1.0 / (1.0 + sqrt(h)) : 1.0; *cs = sc; }
      const double d2 = fmin(fmax(sc * sc * h, min_lm_diagonal), max_lm_diagonal);
      d = d2 / (sc * sc);
      ve = g / d;
    }
    vS[cd] = ve;
    bimg[BI_DH2 + cd] = d;
    bimg[BI_V + cd] = ve;
    part_gn = g * ve;
    part_gmax = fabs(g);
  }
  // v of the pose part (k_assemble_pose wrote it) for the coupling rows' share of q
  if (tid < 80) vS[tid] = bimg[BI_V + tid];
  if (tid < CD_N) act[tid] = cd_active(tid, F, cmask) ? 1 : 0;
  // what this thread's three entries per frame read and write does not depend on the frame: indices worked out once
  //   kind 0 A_kk, 1 A_{k+1,k}^T, 2 coupling row, 3 nothing, 4 a zero-padding row;  si / sj: entry of Gi / Gj (-1: none);  d1 / d2: dimensions
  //   (within their frame's 13) whose activity masks the entry and whose v multiply it in q;  dst: offset inside the frame's block of the target
  //   array;  df, c: a coupling entry's pose (k - 1 + df) and its dimension
  int e_kind[3], e_si[3], e_sj[3], e_d1[3], e_d2[3], e_dst[3], e_df[3], e_c[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int e = tid + AF_THREADS * u;
    e_kind[u] = 3; e_si[u] = -1; e_sj[u] = -1; e_d1[u] = 0; e_d2[u] = 0; e_dst[u] = 0; e_df[u] = 1; e_c[u] = 0;
    if (e < 169) {
      const int i = e / 13, j = e - 13 * i;
      e_kind[u] = 0; e_d1[u] = i; e_d2[u] = j; e_dst[u] = e;
      e_si[u] = tri39(6 + min(i, j), 6 + max(i, j)); e_sj[u] = tri39(25 + min(i, j), 25 + max(i, j));
    } else if (e < 338) {
      const int ji = e - 169, j = ji / 13, i = ji - 13 * j;
      e_kind[u] = 1; e_d1[u] = i; e_d2[u] = j; e_dst[u] = ji;   // d1: dimension of frame k + 1, d2: of frame k
      e_si[u] = tri39(6 + j, 25 + i);
    } else if (e < 626) {
      const int is = e - 338, i = is / 18, sx = is - 18 * i, df = sx / 6, c = sx - 6 * df;
      e_kind[u] = (i < 13) ? 2 : 4;
      e_d1[u] = min(i, 12); e_dst[u] = is; e_df[u] = df; e_c[u] = c;
      if (df == 1) { e_si[u] = tri39(c, 6 + min(i, 12)); e_sj[u] = tri39(19 + c, 25 + min(i, 12)); }
      else if (df == 2) e_si[u] = tri39(6 + min(i, 12), 19 + c);
      else e_sj[u] = tri39(c, 25 + min(i, 12));
    }
  }
  // the prior's diagonal block of frame kb (169 entries, threads 0 .. 168) comes in before the loop
  const double prior_ad = (kb >= 0 && tid < 169) ? pd[PD_AD + 169 * kb + tid] : 0.0;
  __syncthreads();
  // ---- IMU factors. Factor k (frames k, k + 1) contributes a packed 39 x 39 Gram [pose_i 6 | speed / leg-bias_i 13 | pose_j 6 | speed /
  //      leg-bias_j 13 | r] (780 entries, contiguous). Frame by frame through a ring of two Grams in LDS: the factor's 780 entries come in
  //      with coalesced loads, every consumer gathers from LDS, and what goes to the solver (A_kk, A_k+1,k^T, the coupling rows) leaves as
  //      coalesced stores. Per frame k: Gi = factor k (frame k is its "i"), Gj = factor k - 1.
  //        [0, 169)    A_kk             = prior (frame kb) + Gi[6 + .][6 + .] + Gj[25 + .][25 + .]
  //        [169, 338)  A_{k+1,k}^T      = Gi[6 + j][25 + i]
  //        [338, 626)  coupling rows    with poses k - 1 (Gj), k (Gi + Gj), k + 1 (Gi); rows 13 .. 15 zero padding
  auto frame = [&](int k, double (&gp)[4]) {
    const bool has_i = k < F - 1, has_j = k >= 1;
    double *Gi = ring + 780 * (k & 1);
    const double *Gj = ring + 780 * ((k + 1) & 1);
    lds_barrier();   // (the readers of this ring slot — frame k - 1's "Gj" of two frames ago — are done)
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = tid + AF_THREADS * u; if (e < 780) Gi[e] = gp[u]; }   // (frame F - 1: nobody reads it)
    lds_barrier();
    prefetch(k + 2, gp);
    const unsigned char *actk = act + CD_B0 + 13 * k;
    const double *vk = vS + CD_B0 + 13 * k;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int kind = e_kind[u];
      if (kind == 0) {
        double v;
        if (!actk[e_d1[u]] || !actk[e_d2[u]]) v = (e_d1[u] == e_d2[u]) ? 1.0 : 0.0;
        else {
          v = (k == kb) ? prior_ad : 0.0;   // (u == 0 for every A_kk entry: prior_ad is this thread's)
          if (has_i) v += Gi[e_si[u]];
          if (has_j) v += Gj[e_sj[u]];
        }
        bimg[BI_AD + 169 * k + e_dst[u]] = v;
        part_q += vk[e_d1[u]] * v * vk[e_d2[u]];
      } else if (kind == 1) {
        if (k < 10) {
          double v = 0.0;
          if (has_i && actk[13 + e_d1[u]] && actk[e_d2[u]]) v = Gi[e_si[u]];
          bimg[BI_AOT + 169 * k + e_dst[u]] = v;
          part_q += 2.0 * vk[13 + e_d1[u]] * v * vk[e_d2[u]];
        }
      } else if (kind == 2 || kind == 4) {
        const int f = k - 1 + e_df[u];
        double v = 0.0;
        if (kind == 2 && f >= 0 && f < F && actk[e_d1[u]]) {
          if (has_i && e_si[u] >= 0) v += Gi[e_si[u]];
          if (has_j && e_sj[u] >= 0) v += Gj[e_sj[u]];
        }
        bimg[BI_BS + 288 * k + e_dst[u]] = v;
        part_q += 2.0 * vk[e_d1[u]] * v * vS[min(max(6 * f + e_c[u], 0), 79)];
      }
    }
  };
  for (int k = 0; k < F; k += 2) {
    frame(k, gpa);
    if (k + 1 < F) frame(k + 1, gpb);
  }
  PCLK(if (tid == 0) b.st[win].phase_clk[43] = clock64());
  // ---- prior rows of the frame whose speed / leg-bias block the prior touches (rows 13 .. 15: zero padding) ----
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int e = tid + AF_THREADS * u;
    const int i = e / 80, p = e - 80 * i;
    double v = 0.0;
    if (kb >= 0 && i < 13 && p < VILO_NPU && act[CD_B0 + 13 * kb + i] && act[p]) {
      v = pd[PD_BP + e];
      part_q += 2.0 * vS[CD_B0 + 13 * kb + i] * v * vS[p];
    }
    bimg[BI_BP + e] = v;
  }
  // ---- a partial window (F < 11) leaves the blocks of the absent frames as identity / zero for the solver's fixed-size loops ----
  if (F < VILO_MAX_FRAMES) {
    for (int e = tid + 169 * F; e < 11 * 169; e += AF_THREADS) { const int ij = e % 169; bimg[BI_AD + e] = (ij / 13 == ij % 13) ? 1.0 : 0.0; }
    for (int e = tid + 169 * max(F - 1, 0); e < 10 * 169; e += AF_THREADS) bimg[BI_AOT + e] = 0.0;
    for (int e = tid + 288 * F; e < 11 * 288; e += AF_THREADS) bimg[BI_BS + e] = 0.0;
  }
  PCLK(if (tid == 0) b.st[win].phase_clk[44] = clock64());
  // sums of |D^-1 g|^2, max |g| and q over the camera dimensions (the landmarks add theirs in the solver): the pose part's share
  // (k_assemble_pose) first, then this kernel's waves in fixed order
  part_q = wave_sum_dpp(part_q); part_gn = wave_sum_dpp(part_gn); part_gmax = wave_max_dpp(part_gmax);
  if ((tid & 63) == 0) { red[tid >> 6] = part_q; red[4 + (tid >> 6)] = part_gn; red[8 + (tid >> 6)] = part_gmax; }
  __syncthreads();
  if (tid == 0) {
    bimg[BI_SCAL + 0] = bimg[BI_SCAL_POSE + 0] + (((red[0] + red[1]) + red[2]) + red[3]);
    bimg[BI_SCAL + 1] = bimg[BI_SCAL_POSE + 1] + (((red[4] + red[5]) + red[6]) + red[7]);
    bimg[BI_SCAL + 2] = fmax(bimg[BI_SCAL_POSE + 2], fmax(fmax(red[8], red[9]), fmax(red[10], red[11])));
  }
  PCLK(if (tid == 0) b.st[win].phase_clk[46] = clock64());
}

// =================================================================================================
// launch
// =================================================================================================
int vilo_launch_assemble_full(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s, const AcceptParams &ap, int which) {
  (void)ctx;
  if (which == 0) hipLaunchKernelGGL(k_assemble_pose, dim3(b.W), dim3(AF_THREADS), 0, s, b, sp.jacobi_scaling, sp.min_lm_diagonal, sp.max_lm_diagonal, ap);
  else hipLaunchKernelGGL(k_assemble_bias, dim3(b.W), dim3(AF_THREADS), 0, s, b, sp.jacobi_scaling, sp.min_lm_diagonal, sp.max_lm_diagonal);
  return VILO_OK;
}
