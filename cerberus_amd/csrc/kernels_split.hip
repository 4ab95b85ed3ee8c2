// Three-stage form of the single-wave solver for full batches: the block-tridiagonal chain of the speed /
// leg-bias part and the back-substitutions of k_solve_wave (kernels_wave.hip) never touch the pose system's 120 accumulator registers,
// yet inside k_solve_wave they run at ONE wave per SIMD because those accumulators + the Schur pass's operands need the whole register
// file. Cut by register footprint instead:
//
//   k_chain      (256 registers, two waves per SIMD)  S_k, L_k, M_k = L_k^-1, T_A(k), T(k) for frames F-1 .. 0, rhs_P -= T_B^T t_g
//                (chain_common.hpp): T_B(k) in MFMA operand order + reduced right-hand side out (Tk), M_k / T_A(k) out (Lk, TAg)
//   k_solve_mid  (kernels_wave.hip: solve_wave_body<true>; 512 registers)  C -= T_B(k)^T T_B(k) from Tk, landmark Schur complement,
//                blocked Cholesky-80, backward solve: y_P out
//   k_backsub    (<= 128 registers, four waves per SIMD)  c = g_B - B y_P, the two block-bidiagonal sweeps, landmark back-substitution,
//                Gauss-Newton norms, dogleg step, candidate state
//
// The arithmetic is k_solve_wave's, statement by statement: the forms agree bitwise (tests/test_solver_forms.py). What can go wrong
// stays in k_solve_wave: a factorisation that fails in k_chain or in the Cholesky-80, or a non-finite Gauss-Newton step found in
// k_backsub, flags the window for the complete single-kernel path in a fourth launch (k_solve_wave with redo_only) that finds the same
// failure, takes DoglegStrategy::ComputeGaussNewtonStep's retry loop from there, and returns at once for every other window.
// SolverState::pad[1] carries the hand-over: 0 chain failed, 1 chain ready, 2 y_P ready, 3 nothing left to do, 4 redo.
#include <type_traits>
#include "chain_common.hpp"

// ---- k_chain LDS (doubles) ----
#define SC_CH 0         // [CH_TOTAL] chain scratch
#define SC_DB 704       // [144] dhat^2 of the speed / leg-bias part
#define SC_GB 848       // [144] its gradient
#define SC_TOTAL 992
// ---- k_backsub LDS (doubles) ----
#define SB_G 0          // [80]
#define SB_DH2 80       // [80]
#define SB_Y 160        // [80]
#define SB_U 240        // [144]
#define SB_YB 384       // [144]
#define SB_M 528        // [176]
#define SB_TA 704       // [176]
#define SB_DEL 880      // [224]
#define SB_TOTAL 1104
#ifndef BS_WPE
#define BS_WPE 4
#endif

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_chain(BatchDev b) {
  __shared__ __attribute__((aligned(16))) double lds[SC_TOTAL];
  const int win = blockIdx.x;
  SolverState &st = b.st[win];
  if (st.done || !st.need_lin) return;
  const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
  const WinMeta wm = b.win[win];
  const int F = wm.n_frames, kb = wm.pad;
  const double *gin = b.cam_gin + (size_t)win * CD_N;
  const double *bimg = b.Bimg + (size_t)win * BI_N;
  double *Mg = b.Lk + (size_t)win * 11 * 169, *TAg = b.TAg + (size_t)win * 11 * 169;
  double *Tout = b.Tk + (size_t)win * TK_N, *vout = Tout + TK_V;
  const double mu = st.mu;
  for (int e = lane; e < 144; e += 64) { lds[SC_DB + e] = bimg[BI_DH2 + CD_B0 + e]; lds[SC_GB + e] = gin[CD_B0 + e]; }
  lds_fence();
  PCLK(if (lane == 0) st.phase_clk[10] = clock64());
  const int fail = chain_to_global(bimg, gin, lds + SC_CH, lds + SC_DB, lds + SC_GB, Mg, TAg, Tout, vout, F, kb, mu, lane);
  if (lane == 0) st.pad[1] = fail ? 0 : 1;
  PCLK(if (lane == 0) st.phase_clk[11] = clock64());
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BS_WPE, BS_WPE))) k_backsub(BatchDev b, SolveParams sp, int debug_redo) {
  __shared__ __attribute__((aligned(16))) double lds[SB_TOTAL];
  const int win = blockIdx.x;
  SolverState &st = b.st[win];
  if (st.done) return;
  const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
  const WinMeta wm = b.win[win];
  const int F = wm.n_frames, L = wm.L, kb = wm.pad, cmask = wm.const_mask;
  double *g = lds + SB_G, *dh2 = lds + SB_DH2, *y = lds + SB_Y;
  double gBr[3], dBr[3], yBr[3];
  double *cam_g = b.cam_g + (size_t)win * CD_N, *cam_dh2 = b.cam_dh2 + (size_t)win * CD_N, *cam_y = b.cam_y + (size_t)win * CD_N;
  if (st.need_lin) {
    const int stage = st.pad[1];
    if (stage != 2) return;   // 3: the middle stage went through the complete path (or gave up); anything else: it never ran
    const double *gin = b.cam_gin + (size_t)win * CD_N;
    const double mu = st.mu;
    {
      const double *bimg0 = b.Bimg + (size_t)win * BI_N;
      for (int cd = lane; cd < 80; cd += 64) { g[cd] = gin[cd]; dh2[cd] = bimg0[BI_DH2 + cd]; y[cd] = cam_y[cd]; }
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int e = min(lane + 64 * m, 143);
        gBr[m] = gin[CD_B0 + e]; dBr[m] = bimg0[BI_DH2 + CD_B0 + e]; yBr[m] = 0.0;
      }
    }
    lds_fence();
    PCLK(if (lane == 0) st.phase_clk[24] = clock64());
    double gnnorm2 = 0.0, gy = 0.0;
      double part_gnn = 0.0, part_gy = 0.0;
      // (pointers of the second half are formed again from an opaque copy of the window index: carried across the Cholesky as SGPR
      // pairs they push its v_readlane broadcasts into spill lanes)
      int win_b = win, lmoff_b = wm.lm_off;
      asm volatile("" : "+s"(win_b), "+s"(lmoff_b));
      const double *bimg = b.Bimg + (size_t)win_b * BI_N;
      const double *wl = b.lm_w + 80 * (size_t)lmoff_b;
      const double *lm_g = b.lm_gbuf[st.cur] + lmoff_b, *lm_dh2 = b.lm_dh2 + lmoff_b, *lm_einv = b.lm_einv + lmoff_b;
      double *lm_y = b.lm_y + lmoff_b;
      {
        double *U = lds + SB_U, *YB = lds + SB_YB;
        // c: the IMU part of B_k spans poses k-1 .. k+1 (one dimension per lane and trip), the prior part frame kb only
        {
          double bsv[3][18];
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const int e = min(lane + 64 * m, 142), k = e / 13;
#pragma unroll
            for (int s = 0; s < 18; ++s) bsv[m][s] = bimg[BI_BS + (16 * k + (e - 13 * k)) * 18 + s];
          }
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const int e = lane + 64 * m, k = min(e, 142) / 13;
            double sacc = gBr[m];
#pragma unroll
            for (int s = 0; s < 18; ++s) sacc -= bsv[m][s] * y[min(max(6 * (k - 1) + s, 0), 79)];   // (blocks outside the window are zero in the image)
            if (e < 143) U[e] = sacc;
          }
        }
        lds_fence();
        if (kb >= 0) {
          // (unrolled: as a rolled loop every one of its 20 dependent-free loads waited out a memory round trip on its own)
          double sacc = 0.0, bpv[20];
#pragma unroll
          for (int u = 0; u < 20; ++u) bpv[u] = (lr < 13 && lk + 4 * u < VILO_NPU) ? bimg[BI_BP + lr * 80 + lk + 4 * u] : 0.0;
#pragma unroll
          for (int u = 0; u < 20; ++u) sacc += bpv[u] * y[min(lk + 4 * u, 79)];
          sacc += __shfl_xor(sacc, 16, 64);
          sacc += __shfl_xor(sacc, 32, 64);
          if (lane < 13) U[13 * kb + lane] -= sacc;
        }
        lds_fence();
        const int row = lr < 13 ? lr : 0;
        // M_k / T_A(k) of the chain (written to global memory by this wave, L2-resident) come back one frame at a time: the next
        // frame's 2 x 169 values are in flight while this frame's are used out of LDS
        double *MB = lds + SB_M, *TB = lds + SB_TA;
        const double *Mg_ = b.Lk + (size_t)win_b * 11 * 169, *TAg_ = b.TAg + (size_t)win_b * 11 * 169;
        int pe[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) pe[m] = min(lane + 64 * m, 168);
        bool mlow[3];   // M_k = L_k^-1 is lower triangular (exact zeros above the diagonal): not fetched
#pragma unroll
        for (int m = 0; m < 3; ++m) mlow[m] = pe[m] / 13 >= pe[m] % 13;
        double pm[3], pt[3];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < 3; ++m) { pm[m] = mlow[m] ? Mg_[(F - 1) * 169 + pe[m]] : 0.0; pt[m] = 0.0; }
        // forward sweep
        double unext = 0.0;   // u_{k+1}[row]
        for (int k = F - 1; k >= 0; --k) {
#pragma unroll
          for (int m = 0; m < 3; ++m)
            if (lane + 64 * m < 169) { MB[pe[m]] = pm[m]; TB[pe[m]] = pt[m]; }
          {
            // next step: M_{k-1}, T_A(k); after the last one the backward sweep's first frame: M_0 (again) and nothing
            const int kn = max(k - 1, 0);
#pragma unroll
            for (int m = 0; m < 3; ++m) { pm[m] = mlow[m] ? Mg_[kn * 169 + pe[m]] : 0.0; pt[m] = TAg_[k * 169 + pe[m]]; }
          }
          lds_fence();
          double s = U[13 * k + row];
          if (k < F - 1) {
#pragma unroll
            for (int q = 0; q < 13; ++q) s -= TB[q * 13 + row] * readlane_d(unext, q);
          }
          double u = 0.0;
#pragma unroll
          for (int q = 0; q < 13; ++q) u += MB[row * 13 + q] * readlane_d(s, q);
          if (lane < 13) U[13 * k + lane] = u;
          unext = u;
        }
        lds_fence();
        // backward sweep (pm holds M_0; T_A(0) does not exist)
        double yprev = 0.0;
        for (int k = 0; k < F; ++k) {
#pragma unroll
          for (int m = 0; m < 3; ++m)
            if (lane + 64 * m < 169) { MB[pe[m]] = pm[m]; TB[pe[m]] = pt[m]; }
          {
            const int kn = min(k + 1, F - 1);
#pragma unroll
            for (int m = 0; m < 3; ++m) { pm[m] = mlow[m] ? Mg_[kn * 169 + pe[m]] : 0.0; pt[m] = TAg_[kn * 169 + pe[m]]; }
          }
          lds_fence();
          double s = U[13 * k + row];
          if (k > 0) {
#pragma unroll
            for (int q = 0; q < 13; ++q) s -= TB[row * 13 + q] * readlane_d(yprev, q);
          }
          double yk = 0.0;
#pragma unroll
          for (int q = 0; q < 13; ++q) yk += MB[q * 13 + row] * readlane_d(s, q);
          if (!cd_active(CD_B0 + 13 * k + row, F, cmask)) yk = 0.0;
          if (lane < 13) YB[13 * k + lane] = yk;
          yprev = yk;
        }
        lds_fence();
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const int e = lane + 64 * m;
          yBr[m] = (e < 13 * F) ? YB[e] : 0.0;
          part_gnn += dBr[m] * yBr[m] * yBr[m];   // (y is zero on inactive dimensions)
          part_gy += gBr[m] * yBr[m];
        }
      }
      PCLK(if (lane == 0) st.phase_clk[25] = clock64());
      // ---- landmarks: y_l = (g_l - w_l^T yP) / (E_l + mu dhat_l^2), all 80 coupling entries of a landmark in flight at once ----
      const unsigned char *lms = b.lm_s + lmoff_b;
      for (int l = lane; l < L; l += 64) {
        const double gl = lm_g[l], ei = lm_einv[l], d2 = lm_dh2[l];
        const int zl = 6 * (int)lms[l];   // rows before the landmark's start frame are structural zeros: not fetched
        double tl = 0.0;
        // (20 coupling entries in flight per lane: the other waves of the SIMD cover the rest of the latency)
#pragma unroll
        for (int a0 = 0; a0 < 80; a0 += 20) {
          double wcol[20];
#pragma unroll
          for (int a = 0; a < 20; ++a) wcol[a] = (a0 + a >= 60 || a0 + a >= zl) ? wl[(size_t)(a0 + a) * L + l] : 0.0;
#pragma unroll
          for (int a = 0; a < 20; ++a)
            if (a0 + a < VILO_NPU) tl += wcol[a] * y[a0 + a];   // y is zero on inactive dimensions
          asm volatile("" : "+v"(tl));
        }
        const double yl = (gl - tl) * ei;
        lm_y[l] = yl;
        part_gnn += d2 * yl * yl;
        part_gy += gl * yl;
      }
      for (int cd = lane; cd < 80; cd += 64) {
        part_gnn += dh2[cd] * y[cd] * y[cd];
        part_gy += g[cd] * y[cd];
      }
      gnnorm2 = wave_sum(part_gnn);
      gy = wave_sum(part_gy);
    PCLK(if (lane == 0) st.phase_clk[26] = clock64());
    if (!(isfinite(gnnorm2) && isfinite(gy))) {   // IsArrayValid(gauss_newton_step_) failed: mu * 10 and the complete path again
      const double mu2 = mu * 10.0;
      if (lane == 0) {
        st.mu = mu2; st.pad[0]++;
        if (!(mu2 < 1.0)) { st.lin_fail = 1; st.step_valid = 0; st.scale_ready = 1; st.pad[1] = 3; }
        else st.pad[1] = 4;
      }
      return;
    }
    if (debug_redo) {   // (tests: send every window through the fourth launch with mu as it is)
      if (lane == 0) st.pad[1] = 4;
      return;
    }
    for (int cd = lane; cd < 80; cd += 64) { cam_g[cd] = g[cd]; cam_dh2[cd] = dh2[cd]; cam_y[cd] = y[cd]; }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int e = lane + 64 * m;
      if (e < 144) { cam_g[CD_B0 + e] = gBr[m]; cam_dh2[CD_B0 + e] = dBr[m]; cam_y[CD_B0 + e] = yBr[m]; }
    }
    if (lane == 0) {
      st.gnnorm2 = gnnorm2; st.gdotgn = -gy;
      st.alpha = st.gnorm2 / st.q;
      st.scale_ready = 1;
      st.lin_fail = 0;
      st.pad[1] = 3;
    }
  } else {
    for (int cd = lane; cd < 80; cd += 64) { g[cd] = cam_g[cd]; dh2[cd] = cam_dh2[cd]; y[cd] = cam_y[cd]; }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int e = min(lane + 64 * m, 143);
      gBr[m] = cam_g[CD_B0 + e]; dBr[m] = cam_dh2[CD_B0 + e]; yBr[m] = cam_y[CD_B0 + e];
    }
  }
  lds_fence();

  // ---- dogleg step for the current radius, candidate camera state ----
  double ca = 0.0, cb = 0.0;
  int go = 0;
  if (lane == 0) {
    if (st.radius <= sp.min_radius) { st.done = 1; st.termination = 1; st.step_valid = 0; }
    else { dogleg_scalars(st); ca = st.coef_a; cb = st.coef_b; go = st.step_valid; }
  }
  ca = readlane_d(ca, 0); cb = readlane_d(cb, 0); go = __builtin_amdgcn_readlane(go, 0);
  int win_d = win;
  asm volatile("" : "+s"(win_d));
  const double *x = b.x + (size_t)win_d * XSTRIDE;
  double *xc = b.xc + (size_t)win_d * XSTRIDE;
  {
    // candidate inverse depths: lambda_c = lambda - a g_l / dhat_l^2 - b y_l (no valid step: the candidate is the current point, which
    // the next pass linearises again — HandleInvalidStep in k_accept)
    const double *lam = b.lam + wm.lm_off, *lmg = b.lm_gbuf[st.cur] + wm.lm_off, *lmd = b.lm_dh2 + wm.lm_off, *lmy = b.lm_y + wm.lm_off;
    double *lamc = b.lamc + wm.lm_off;
    for (int l = lane; l < L; l += 64) lamc[l] = go ? lam[l] - ca * lmg[l] / lmd[l] - cb * lmy[l] : lam[l];
  }
  if (!go) {
    for (int e = lane; e < XSTRIDE; e += 64) xc[e] = x[e];
    return;
  }
  double *del = lds + SB_DEL;
  for (int cd = lane; cd < 80; cd += 64) del[cd] = -ca * g[cd] / dh2[cd] - cb * y[cd];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int e = lane + 64 * m;
    if (e < 143) del[CD_B0 + e] = -ca * gBr[m] / dBr[m] - cb * yBr[m];
  }
  lds_fence();
  if (lane < 11) pose_plus(x + XO_POSE + 7 * lane, del + 6 * lane, xc + XO_POSE + 7 * lane);
  else if (lane < 13) pose_plus(x + XO_EX + 7 * (lane - 11), del + CD_EX0 + 6 * (lane - 11), xc + XO_EX + 7 * (lane - 11));
  else if (lane == 13) xc[XO_TD] = x[XO_TD] + del[CD_TD];
  for (int e = lane; e < 143; e += 64) {
    const int k = e / 13, c = e - 13 * k;
    if (c < 9) xc[XO_SB + 9 * k + c] = x[XO_SB + 9 * k + c] + del[CD_B0 + e];
    else xc[XO_LB + 4 * k + (c - 9)] = x[XO_LB + 4 * k + (c - 9)] + del[CD_B0 + e];
  }
  PCLK(if (lane == 0) st.phase_clk[27] = clock64());
}

int vilo_launch_split_stage(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s, int which) {
  (void)ctx;
  if (which == 0) hipLaunchKernelGGL(k_chain, dim3(b.W), dim3(64), 0, s, b);
  else {
    static const int debug_redo = [] { const char *e = getenv("VILO_DEBUG_REDO"); return e ? atoi(e) : 0; }();
    hipLaunchKernelGGL(k_backsub, dim3(b.W), dim3(64), 0, s, b, sp, debug_redo);
  }
  return VILO_OK;
}
