// Multi-wave solver for gfx950: the same algorithm as k_solve_wave (kernels_wave.hip: what ceres::Solve does per linearisation for
// Estimator::optimization(), estimator.cpp:1221-1236 — DENSE_SCHUR + traditional DOGLEG, Ceres 1.14 semantics) with one WORKGROUP OF TWO
// WAVES per window instead of one wave, because the single wave holds two unrelated register footprints that never overlap in time:
//
//   wave A ("chain")   the block-tridiagonal Cholesky of the speed / leg-bias part (frames F-1 .. 0: 13 x 13 factorisation and substitutions
//                      with lane = row, v_readlane pivots) with its coupling rows T(k) (13 x 80 per frame, FP64-MFMA tiles), later the
//                      speed / leg-bias back-substitution sweeps;
//   wave B ("matrix")  the 80 x 80 pose system in 120 accumulator registers: landmark Schur complement straight from global memory,
//                      C -= T(k)^T T(k) with T(k) handed over through LDS (double-buffered, one s_barrier per frame), blocked Cholesky
//                      with the forward solve riding along, backward solve, landmark back-substitution.
//
// The landmark Schur complement does not depend on the chain, so the two run side by side: a frame of the chain (about 12 k cycles of
// dependent scalar work) against one trip of 16 landmarks + the rank update of the previous frame (about 11 k cycles of MFMA issue).
// A window's solve takes about 0.6 of the single wave's time this way. The pair holds two SIMDs (the 120 accumulators of wave B do not
// leave room for a second wave on its SIMD), so on a full chip the single-wave form still moves more windows per second: this form is the
// one for batches that leave SIMDs idle (up to two windows per CU: BASELINE configs[3] is 128 windows per GPU, a robot is one), chosen
// by batch size like the frame-parallel linearisation. Results agree with k_solve_wave to rounding (the landmark and norm sums are split
// over two waves), batch-of-N == batch-of-1 bitwise as before.
#include <type_traits>
#include "wave_common.hpp"

// LDS map (doubles)
#define MW_G 0          // [80]  gradient of the pose part
#define MW_DH2 80       // [80]  dogleg diagonal
#define MW_Y 160        // [80]  Gauss-Newton step of the pose part
#define MW_V 240        // [80]  reduced right-hand side
#define MW_DB 320       // [144] dogleg diagonal of the speed / leg-bias part
#define MW_GB 464       // [144] its gradient
#define MW_RED 608      // [32]  cross-wave sums and flags
#define MW_CH 640       // [704] chain scratch: M_k, T_A(k) x 2, S_{k-1};  later the sweeps' U, YB, M, T_A
#define MW_T 1344       // [2560] T(k) hand-over, two buffers of 5 tiles x 4 registers x 64 lanes;  later the Cholesky's scratch and panel slots, then the step
#define MW_SKIP 3904    // [40]  per-trip flags of the Schur pass (ints)
#define MW_VP 3944      // [80]  v_P = D^-2 g of the pose part (cross term of q)
#define MW_TOTAL 4024
// chain scratch
#define MC_LM 0
#define MC_TA0 176
#define MC_TA1 352
#define MC_SN 528
// sweeps (same region)
#define MB_U 0
#define MB_YB 144
#define MB_M 288
#define MB_TA 464
// Cholesky (MW_T region)
#define MX_D16 0
#define MX_LI16 272
#define MX_P16 560
#define MX_PANEL 1024   // 4 slots x 256
#define MX_DEL 2048     // [224] step of the camera dimensions (after the Cholesky)
// reduction slots
#define MR_GN 0         // [2] |D^-1 g|^2 parts
#define MR_GMAX 2
#define MR_Q 4
#define MR_FAIL 6       // [2] factorisation failed (chain / pose system)
#define MR_GNN 8
#define MR_GY 10
#define MR_CA 12        // dogleg coefficients a, b, valid flag
#define MR_CB 13
#define MR_GO 14
#define MR_QX 15        // cross term of q from the Schur pass

extern "C" size_t vilo_solve_mw_lds_bytes() { return (size_t)MW_TOTAL * sizeof(double); }

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// global writes of one wave read by the other (same CU): drain them, then meet
__device__ __forceinline__ void wg_barrier_global() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ void __launch_bounds__(128) k_solve_mw(BatchDev b, SolveParams sp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int win = blockIdx.x;
  SolverState &st = b.st[win];
  if (st.done) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int tid = threadIdx.x;
  const WinMeta wm = b.win[win];
  const int F = wm.n_frames, L = wm.L, kb = wm.pad, cmask = wm.const_mask;
  double *g = lds + MW_G, *dh2 = lds + MW_DH2, *y = lds + MW_Y, *v = lds + MW_V, *red = lds + MW_RED;
  // wave A: gradient, dogleg diagonal and Gauss-Newton step of the 143 speed / leg-bias dimensions: dimension lane + 64 m in register m
  double gBr[3] = {0.0, 0.0, 0.0}, dBr[3] = {1.0, 1.0, 1.0}, yBr[3] = {0.0, 0.0, 0.0};

  if (st.need_lin) {
    const double *bimg = b.Bimg + (size_t)win * BI_N;
    const double *gin = b.cam_gin + (size_t)win * CD_N;
    const double *wl = b.lm_w + 80 * (size_t)wm.lm_off;
    double *lm_E = b.lm_E + wm.lm_off, *lm_g = b.lm_gbuf[st.cur] + wm.lm_off, *lm_dh2 = b.lm_dh2 + wm.lm_off, *lm_scale = b.lm_scale + wm.lm_off,
           *lm_einv = b.lm_einv + wm.lm_off, *lm_y = b.lm_y + wm.lm_off;
    const bool first_scale = !st.scale_ready;
    double mu = st.mu;

    PCLK(if (tid == 0) st.phase_clk[0] = clock64());
    // ---- camera-side vectors come scaled from k_assemble; the landmarks are scaled here (both waves, landmark tid + 128 n) ----
    double part_gn = 0.0, part_gmax = 0.0, part_q = 0.0;
    if (wave == 1) {
      for (int cd = lane; cd < 80; cd += 64) { g[cd] = gin[cd]; dh2[cd] = bimg[BI_DH2 + cd]; lds[MW_VP + cd] = cd_active(cd, F, cmask) ? bimg[BI_V + cd] : 0.0; }
    } else {
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int e = min(lane + 64 * m, 143);
        gBr[m] = gin[CD_B0 + e]; dBr[m] = bimg[BI_DH2 + CD_B0 + e];
        if (lane + 64 * m < 144) { lds[MW_DB + e] = dBr[m]; lds[MW_GB + e] = gBr[m]; }
      }
    }
    for (int l = tid; l < L; l += 128) {
      const double E = lm_E[l], gl = lm_g[l];
      double sc;
      if (first_scale) { sc = sp.jacobi_scaling ? 1.0 / (1.0 + sqrt(E)) : 1.0; lm_scale[l] = sc; }
      else sc = lm_scale[l];
      const double d2 = fmin(fmax(sc * sc * E, sp.min_lm_diagonal), sp.max_lm_diagonal) / (sc * sc);
      lm_dh2[l] = d2;
      const double vl = gl / d2;
      part_q += E * vl * vl;   // the cross term 2 vl w_l^T v is accumulated in the Schur pass
      lm_y[l] = vl;            // (scratch until the back-substitution overwrites it)
      part_gn += gl * vl;
      part_gmax = fmax(part_gmax, fabs(gl));
    }
    part_gn = wave_sum(part_gn); part_gmax = wave_max(part_gmax); part_q = wave_sum(part_q);
    if (lane == 0) { red[MR_GN + wave] = part_gn; red[MR_GMAX + wave] = part_gmax; red[MR_Q + wave] = part_q; }
    wg_barrier_global();
    const double gnorm2 = bimg[BI_SCAL + 1] + (red[MR_GN] + red[MR_GN + 1]), gmax = fmax(bimg[BI_SCAL + 2], fmax(red[MR_GMAX], red[MR_GMAX + 1]));
    const double q_lm = red[MR_Q] + red[MR_Q + 1];
    PCLK(if (tid == 0) st.phase_clk[1] = clock64());
    if (!sp.fixed_iterations && gmax <= sp.gradient_tolerance) {
      if (tid == 0) { st.gmax = gmax; st.done = 1; st.termination = 1; st.step_valid = 0; }
      return;
    }

    bool solved = false;
    double qq = 0.0, gnnorm2 = 0.0, gy = 0.0;
    const int lane_outer = lane;
    while (!solved) {
      // (the body repeats only when a factorisation fails and mu grows; the lane index is made opaque per trip so that lane-derived
      // addresses are not hoisted out of the "loop" and kept alive across the whole kernel)
      int lane = lane_outer;
      asm volatile("" : "+v"(lane));
      const int lr = lane & 15, lk = lane >> 4;
      for (int l = tid; l < L; l += 128) lm_einv[l] = 1.0 / (lm_E[l] + mu * lm_dh2[l]);
      wg_barrier_global();   // (lm_einv, lm_y, lm_dh2 are read back through global memory by lanes of both waves)
      PCLK(if (tid == 0) st.phase_clk[2] = clock64());
      const int nks = (L + 3) >> 2, NT = (nks + 3) >> 2;   // k-steps of 4 landmarks, trips of 4 k-steps
      int fail = 0;

      if (wave == 0) {
        // =============================== wave A: the chain ===============================
        //   S_k = A_kk + mu D_k - T_A(k+1)^T T_A(k+1),  L_k = chol(S_k),  M_k = L_k^-1,  T_A(k) = M_k A_{k,k-1},
        //   V = [B_k | g_k] - T_A(k+1)^T T(k+1),  T(k) = M_k V   (13 x 80: columns 0..78 coupling rows, column 79 = rhs)
        // T(k) goes to wave B through LDS in accumulator order (register r of lane (lr, lk) = row lk + 4 r, column 16 X + lr): that is
        // the operand layout of C -= T^T T there and of V -= T_A^T T here.
        double *Mg = b.Lk + (size_t)win * 11 * 169, *TAg = b.TAg + (size_t)win * 11 * 169;
        int fX[5], oX[5];
#pragma unroll
        for (int X = 0; X < 5; ++X) { const int col = 16 * X + lr; fX[X] = col < 66 ? col / 6 : 99; oX[X] = col < 66 ? col - 6 * fX[X] : 0; }
        const int grp = lk, c = lr;
        const int row = c < 13 ? c : 0;
        double *scr = lds + MW_CH;
        double *LM = scr + MC_LM, *SN = scr + MC_SN;
        double *TAcur = scr + MC_TA0, *TAprev = scr + MC_TA1;
        const double *DB = lds + MW_DB, *GB = lds + MW_GB;
        mfma_d4 T[5];
#pragma unroll
        for (int X = 0; X < 5; ++X) T[X] = mfma_d4{0.0, 0.0, 0.0, 0.0};
        auto load_blocks = [&](int k, mfma_d4 *Vn, double *rhsn, double *adnn) {
#pragma unroll
          for (int X = 0; X < 5; ++X) {
            const int df = fX[X] - k + 1;
            const bool on = df >= 0 && df <= 2;
            const double *src = bimg + BI_BS + (k * 16 + lk) * 18 + 6 * min(max(df, 0), 2) + oX[X];
#pragma unroll
            for (int r = 0; r < 4; ++r) Vn[X][r] = on ? src[72 * r] : 0.0;
          }
          if (k == kb) {
#pragma unroll
            for (int X = 0; X < 5; ++X)
#pragma unroll
              for (int r = 0; r < 4; ++r) Vn[X][r] += bimg[BI_BP + (lk + 4 * r) * 80 + 16 * X + lr];
          }
#pragma unroll
          for (int i = 0; i < 13; ++i) rhsn[i] = (k > 0) ? bimg[BI_AOT + (max(k - 1, 0) * 13 + row) * 13 + i] : 0.0;   // column `row` of A_{k,k-1}
#pragma unroll
          for (int r = 0; r < 4; ++r) adnn[r] = (k > 0 && lr < 13 && lk + 4 * r < 13) ? bimg[BI_AD + max(k - 1, 0) * 169 + (lk + 4 * r) * 13 + lr] : 0.0;
        };
        PCLK(if (lane == 0) st.phase_clk[16] = clock64());
        mfma_d4 nV[5];
        double nrhs[13], nadn[4];
        load_blocks(F - 1, nV, nrhs, nadn);
        for (int i = 0; i <= F; ++i) {
          if (i < F) {
            const int k = F - 1 - i;
            const int x_lo = (k <= kb) ? 0 : max(0, (6 * (k - 1)) >> 4);   // T(k) is zero left of pose k - 1 (dense from the prior's frame down)
            mfma_d4 V[5];
            double a[13], l[13], rhs[13], adn[4];
#pragma unroll
            for (int X = 0; X < 5; ++X) V[X] = nV[X];
#pragma unroll
            for (int q = 0; q < 13; ++q) rhs[q] = nrhs[q];
#pragma unroll
            for (int m = 0; m < 4; ++m) adn[m] = nadn[m];
            if (lr == 15) {
#pragma unroll
              for (int r = 0; r < 4; ++r) V[4][r] = (lk + 4 * r < 13) ? GB[13 * k + lk + 4 * r] : 0.0;
            }
            if (k == F - 1) {
#pragma unroll
              for (int j = 0; j < 13; ++j) a[j] = bimg[BI_AD + (k * 13 + row) * 13 + j];
            } else {
#pragma unroll
              for (int j = 0; j < 13; ++j) a[j] = SN[row * 13 + j];
            }
            {
              const double md = mu * DB[13 * k + row];
#pragma unroll
              for (int j = 0; j < 13; ++j) a[j] += (j == row) ? md : 0.0;
            }
            double myrinv = 1.0;
#pragma unroll
            for (int j = 0; j < 13; ++j) {
              double piv = readlane_d(a[j], j);
              if (!(piv > 0.0) || !isfinite(piv)) { fail = 1; piv = 1.0; }
              const double rinv = rsqrt(piv);
              const double lj = (c == j) ? piv * rinv : (c > j ? a[j] * rinv : 0.0);
              l[j] = lj;
              if (c == j) myrinv = rinv;
#pragma unroll
              for (int q = j + 1; q < 13; ++q) a[q] -= lj * readlane_d(lj, q);
            }
            // forward substitutions L x = rhs: T_A(k) columns (group 0), L^-1 columns (group 1); L broadcast from the owning lanes
#pragma unroll
            for (int j = 0; j < 13; ++j) asm volatile("" : "+v"(l[j]));
            double cl[13];
#pragma unroll
            for (int q2 = 0; q2 < 13; ++q2) {
              double vv = (grp == 0) ? rhs[q2] : ((q2 == c) ? 1.0 : 0.0);
#pragma unroll
              for (int q = 0; q < q2; ++q) vv -= readlane_d(l[q], q2) * cl[q];
              cl[q2] = vv * readlane_d(myrinv, q2);
              asm volatile("" : "+v"(cl[q2]));
              __builtin_amdgcn_sched_barrier(0);   // (one row's v_readlane results at a time)
            }
            if (c < 13 && grp < 2) {
              if (grp == 0) {
#pragma unroll
                for (int q = 0; q < 13; ++q) { TAcur[q * 13 + c] = cl[q]; TAg[k * 169 + q * 13 + c] = cl[q]; }
              } else {
#pragma unroll
                for (int q = 0; q < 13; ++q) { LM[q * 13 + c] = cl[q]; Mg[k * 169 + q * 13 + c] = cl[q]; }
              }
            }
            lds_fence();
            // S_{k-1} = A_{k-1,k-1} - T_A(k)^T T_A(k): one 16 x 16 tile on the matrix cores
            if (k > 0) {
              mfma_d4 sn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) {
                const int q = 4 * kk + lk;
                const double ta = ((lr < 13) && (q < 13)) ? TAcur[min(q, 12) * 13 + min(lr, 12)] : 0.0;
                sn = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, ta, sn, 0, 0, 0);
              }
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (lr < 13 && lk + 4 * r < 13) SN[(lk + 4 * r) * 13 + lr] = adn[r] - sn[r];
            }
            // V -= T_A(k+1)^T T(k+1);  T(k) = M_k V
            double at[4], am[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const int q = 4 * kk + lk;
              const bool in = (lr < 13) && (q < 13);
              const double ta = TAprev[min(q, 12) * 13 + min(lr, 12)], m = LM[min(lr, 12) * 13 + min(q, 12)];
              at[kk] = (in && k < F - 1) ? -ta : 0.0;
              am[kk] = in ? m : 0.0;
            }
            if (k < F - 1) {
#pragma unroll
              for (int X = 0; X < 5; ++X)
                if (X >= x_lo) {
#pragma unroll
                  for (int kk = 0; kk < 4; ++kk) V[X] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[kk], T[X][kk], V[X], 0, 0, 0);
                }
            }
#pragma unroll
            for (int X = 0; X < 5; ++X)
              if (X >= x_lo) {
                mfma_d4 n = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) n = __builtin_amdgcn_mfma_f64_16x16x4f64(am[kk], V[X][kk], n, 0, 0, 0);
                T[X] = n;
              }
            // the next frame's blocks: in flight behind the hand-over
            if (k > 0) load_blocks(k - 1, nV, nrhs, nadn);
            // hand T(k) over (tiles left of x_lo are zero and skipped on both sides)
            double *Tb = lds + MW_T + 1280 * (i & 1);
#pragma unroll
            for (int X = 0; X < 5; ++X)
              if (X >= x_lo) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) Tb[(X * 4 + kk) * 64 + lane] = T[X][kk];
              }
            double *sw = TAcur; TAcur = TAprev; TAprev = sw;
          }
          PCLK(if (i == F - 1 && lane == 0) st.phase_clk[17] = clock64());
          if (i == F && lane == 0) red[MR_FAIL] = (double)fail;
          wg_barrier();
        }
      } else {
        // =============================== wave B: the pose system ===============================
        const double *Cimg = b.Cimg + (size_t)win * CIMG_N;
        mfma_d4 acc[15];
#pragma unroll
        for (int t = 0; t < 15; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] = Cimg[(t * 4 + r) * 64 + lane];
        // ---- Schur complement of the landmarks on the FP64 matrix cores, C -= sum_l w_l w_l^T / (E_l + mu dhat_l^2), interleaved with the
        //      rank updates C -= T(k)^T T(k) of the chain's frames as wave A delivers them: step i = a share of the landmark trips, then
        //      the frame handed over in step i - 1. One k-step = 4 landmarks; the operand of tile row X (lane: w[16 X + lr][4 kk + lk])
        //      serves as A of tiles (X, .) and as B of tiles (., X): 5 row-coalesced global loads and 15 MFMAs per k-step, no LDS. ----
        // (the coupling rows of constant blocks are written as zeros by the solve passes of the visual kernels: no masks here; the cross
        // term of q is formed by the landmark back-substitution, which reads every coupling entry anyway)
        double yacc[5];
#pragma unroll
        for (int X = 0; X < 5; ++X) yacc[X] = 0.0;
        // A landmark that starts in frame s couples with poses s .. only, and the landmarks of a window are ordered by start frame: a trip of
        // 16 landmarks whose first one starts in frame s >= 3 has nothing in tile row 0 (10 tiles instead of 15; two branch-free forms)
        int *skip_tab = (int *)(lds + MW_SKIP);
        if (L > 0) {
          const unsigned char *lms = b.lm_s + wm.lm_off;
          for (int tr = lane; tr < NT + 2; tr += 64) skip_tab[tr] = (6 * (int)lms[min(16 * tr, L - 1)] >= 16) ? 1 : 0;
          lds_fence();
        }
        // half trips of 2 k-steps = 8 landmarks: the operands of the next half are in flight behind the 30 MFMAs of this one
        double opb[2][2][5], eb[2][2], gb[2][2];
        auto ldhalf = [&](int h, auto bs) {
          constexpr int bsel = decltype(bs)::value;
          const int kk0 = 2 * h;
          const int skip0 = __builtin_amdgcn_readfirstlane(skip_tab[min(h >> 1, NT)]);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int l = 4 * (kk0 + u) + lk, lc = min(l, L - 1);
            eb[bsel][u] = (l < L) ? lm_einv[lc] : 0.0; gb[bsel][u] = lm_g[lc];
#pragma unroll
            for (int X = 1; X < 5; ++X) opb[bsel][u][X] = wl[(size_t)(16 * X + lr) * L + lc];
          }
          if (!skip0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) opb[bsel][u][0] = wl[(size_t)lr * L + min(4 * (kk0 + u) + lk, L - 1)];
          }
        };
        auto dohalf = [&](auto bs, auto xl) {
          constexpr int bsel = decltype(bs)::value;
          constexpr int XL = decltype(xl)::value;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const double ei = eb[bsel][u], ge = gb[bsel][u] * ei;
#pragma unroll
            for (int t = 0; t < 15; ++t)
              if (c_tJ[t] >= XL) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-(opb[bsel][u][c_tI[t]] * ei), opb[bsel][u][c_tJ[t]], acc[t], 0, 0, 0);
#pragma unroll
            for (int X = XL; X < 5; ++X) yacc[X] += opb[bsel][u][X] * ge;
          }
        };
        auto half = [&](int h, auto bs) {
          const int skip0 = __builtin_amdgcn_readfirstlane(skip_tab[min(h >> 1, NT)]);
          if (skip0) dohalf(bs, std::integral_constant<int, 1>{});
          else dohalf(bs, std::integral_constant<int, 0>{});
        };
        double yr[5];
#pragma unroll
        for (int X = 0; X < 5; ++X) yr[X] = 0.0;
        const int NH = (nks + 1) >> 1;   // half trips
        if (NH > 0) ldhalf(0, std::integral_constant<int, 0>{});
        int h = 0;
        for (int i = 0; i <= F; ++i) {
          const int h_end = ((i + 1) * NH) / (F + 1);
          for (; h < h_end; ++h) {
            // (the buffer parity is a wave-uniform branch)
            if (h & 1) {
              if (h + 1 < NH) ldhalf(h + 1, std::integral_constant<int, 0>{});
              half(h, std::integral_constant<int, 1>{});
            } else {
              if (h + 1 < NH) ldhalf(h + 1, std::integral_constant<int, 1>{});
              half(h, std::integral_constant<int, 0>{});
            }
          }
          if (i >= 1) {
            // C -= T_B(k)^T T_B(k), rhs_P -= T_B(k)^T t_g(k) for the frame wave A finished in step i - 1; one k-step (4 rows of T) at a time
            const int k = F - i;
            const int x_lo = (k <= kb) ? 0 : max(0, (6 * (k - 1)) >> 4);
            const double *Tb = lds + MW_T + 1280 * ((i - 1) & 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              double Tk[5];
#pragma unroll
              for (int X = 0; X < 5; ++X) Tk[X] = (X >= x_lo) ? Tb[(X * 4 + kk) * 64 + lane] : 0.0;
              // t_g(k) (column 79) to every lane of its 16-lane row group; the pose system must not see it
              const double tg = __shfl(Tk[4], (lane & 48) | 15, 64);
              if (lr == 15) Tk[4] = 0.0;
#pragma unroll
              for (int t = 0; t < 15; ++t)
                if (c_tJ[t] >= x_lo) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Tk[c_tI[t]], Tk[c_tJ[t]], acc[t], 0, 0, 0);
#pragma unroll
              for (int X = 0; X < 5; ++X) yr[X] += Tk[X] * tg;
            }
          }
          wg_barrier();
        }
        PCLK(if (lane == 0) st.phase_clk[3] = clock64());
        fail = (red[MR_FAIL] != 0.0) ? 1 : 0;
        // reduced right-hand side: g_P - sum_k T_B^T t_g - sum_l w_l g_l / (E_l + mu dhat_l^2)
#pragma unroll
        for (int X = 0; X < 5; ++X) {
          double s = yr[X] + yacc[X];
          s += __shfl_xor(s, 16, 64);
          s += __shfl_xor(s, 32, 64);
          if (lk == 0) v[16 * X + lr] = g[16 * X + lr] - s;
        }
        // regularise: diag += mu dhat^2
#pragma unroll
        for (int I = 0; I < 5; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (lk + 4 * r == lr) acc[tile_index(I, I)][r] += mu * dh2[16 * I + lr];
        lds_fence();

        // ---- dense Cholesky of the 80 x 80 reduced pose system, blocked by 16 (as in k_solve_wave): diagonal tile in registers + v_readlane
        //      (also its inverse), panel and trailing update on the matrix cores, the right-hand side riding along as a sixth block row;
        //      the factor stays in the accumulator registers = the operand order of L^T for the backward solve ----
        if (!fail) {
          double *scr = lds + MW_T;
          double *D16 = scr + MX_D16, *LI16 = scr + MX_LI16, *P16 = scr + MX_P16;
          double vrow[5];
#pragma unroll
          for (int J = 0; J < 5; ++J) vrow[J] = (lk == 0) ? v[16 * J + lr] : 0.0;
#pragma unroll
          for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) D16[(lk + 4 * r) * 17 + lr] = acc[tile_index(j, j)][r];
            lds_fence();
            fail |= chol16_tile(D16, LI16);
            double li[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) li[kk] = LI16[lr * 17 + 4 * kk + lk];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tile_index(j, j)][r] = LI16[(lk + 4 * r) * 17 + lr];   // L_jj^-1 in accumulator order
#pragma unroll
            for (int I = j + 1; I < 5; ++I) {
              const int t = tile_index(I, j);
#pragma unroll
              for (int r = 0; r < 4; ++r) P16[(lk + 4 * r) * 17 + lr] = acc[t][r];
              lds_fence();
              mfma_d4 nacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) nacc = __builtin_amdgcn_mfma_f64_16x16x4f64(P16[lr * 17 + 4 * kk + lk], li[kk], nacc, 0, 0, 0);
              acc[t] = nacc;
#pragma unroll
              for (int r = 0; r < 4; ++r) lds[pswz_at(MW_T + MX_PANEL, I - j - 1, lk + 4 * r, lr)] = nacc[r];
              lds_fence();   // (P16 is reused by the next panel)
            }
            double pv[4];
            {
#pragma unroll
              for (int r = 0; r < 4; ++r) P16[(lk + 4 * r) * 17 + lr] = (r == 0) ? vrow[j] : 0.0;
              lds_fence();
              mfma_d4 nacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) nacc = __builtin_amdgcn_mfma_f64_16x16x4f64(P16[lr * 17 + 4 * kk + lk], li[kk], nacc, 0, 0, 0);
              if (lk == 0) y[16 * j + lr] = nacc[0];
              lds_fence();
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) pv[kk] = (lr == 0) ? y[16 * j + 4 * kk + lk] : 0.0;
            }
            double pa[5][4];
#pragma unroll
            for (int I = j + 1; I < 5; ++I)
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) pa[I][kk] = lds[pswz_at(MW_T + MX_PANEL, I - j - 1, lr, 4 * kk + lk)];
#pragma unroll
            for (int I = j + 1; I < 5; ++I)
#pragma unroll
              for (int J = j + 1; J <= I; ++J)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                  acc[tile_index(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[I][kk], pa[J][kk], acc[tile_index(I, J)], 0, 0, 0);
#pragma unroll
            for (int J = j + 1; J < 5; ++J) {
              mfma_d4 tv = {vrow[J], 0.0, 0.0, 0.0};
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) tv = __builtin_amdgcn_mfma_f64_16x16x4f64(-pv[kk], pa[J][kk], tv, 0, 0, 0);
              vrow[J] = tv[0];
            }
            lds_fence();   // (the panel slots are rewritten by the next block column)
          }
        }
        PCLK(if (lane == 0) st.phase_clk[4] = clock64());
        if (!fail) {
          // ---- L^T yP = y, blockwise on the matrix cores: x_j = L_jj^-T (y_j - sum_{i>j} L_ij^T x_i); the A operands are the factor's
          //      accumulator registers as they stand ----
          mfma_d4 yb[5];
#pragma unroll
          for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) yb[j][r] = y[16 * j + lk + 4 * r];
#pragma unroll
          for (int j = 4; j >= 0; --j) {
            mfma_d4 accv = yb[j];
#pragma unroll
            for (int i2 = j + 1; i2 < 5; ++i2)
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) accv = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[tile_index(i2, j)][kk], yb[i2][kk], accv, 0, 0, 0);
            mfma_d4 n = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) n = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[tile_index(j, j)][kk], accv[kk], n, 0, 0, 0);
            yb[j] = n;
          }
          lds_fence();
          if (lr == 0) {
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) y[16 * j + lk + 4 * r] = cd_active(16 * j + lk + 4 * r, F, cmask) ? yb[j][r] : 0.0;
          }
        }
        PCLK(if (lane == 0) st.phase_clk[5] = clock64());
        if (lane == 0) red[MR_FAIL + 1] = (double)fail;
      }
      wg_barrier();   // y_P (or the failure flags) visible to both waves
      PCLK(if (lane == 0) st.phase_clk[wave ? 6 : 18] = clock64());
      fail = (red[MR_FAIL] != 0.0 || red[MR_FAIL + 1] != 0.0) ? 1 : 0;
      if (fail) {
        // DoglegStrategy::ComputeGaussNewtonStep: mu *= 10 and retry while mu < max_mu (1.0)
        mu *= 10.0;
        if (tid == 0) { st.mu = mu; st.pad[0]++; }   // (pad[0]: factorisation retries of this solve, read by the tests)
        if (!(mu < 1.0)) {
          if (tid == 0) { st.lin_fail = 1; st.step_valid = 0; st.gnorm2 = gnorm2; st.q = 0.0; st.gmax = gmax; st.scale_ready = 1; }
          return;
        }
        wg_barrier();   // (every lane has read the flags before the next trip rewrites them)
        continue;
      }

      // ---- back-substitution: wave A the speed / leg-bias part, wave B the landmarks ----
      double part_gnn = 0.0, part_gy = 0.0, part_qx = 0.0;
      if (wave == 0) {
        //   c = g_B - B yP, then the two block-bidiagonal sweeps
        //   u_k = M_k (c_k - T_A(k+1)^T u_{k+1})   k = F-1 .. 0,      y_k = M_k^T (u_k - T_A(k) y_{k-1})   k = 0 .. F-1
        double *U = lds + MW_CH + MB_U, *YB = lds + MW_CH + MB_YB;
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
        PCLK(if (lane == 0) st.phase_clk[19] = clock64());
        const int row = lr < 13 ? lr : 0;
        // M_k / T_A(k) of the chain (written to global memory by this wave, L2-resident) come back one frame at a time
        double *MB = lds + MW_CH + MB_M, *TB = lds + MW_CH + MB_TA;
        const double *Mg_ = b.Lk + (size_t)win * 11 * 169, *TAg_ = b.TAg + (size_t)win * 11 * 169;
        int pe[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) pe[m] = min(lane + 64 * m, 168);
        double pm[3], pt[3];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < 3; ++m) { pm[m] = Mg_[(F - 1) * 169 + pe[m]]; pt[m] = 0.0; }
        double unext = 0.0;   // u_{k+1}[row]
        for (int k = F - 1; k >= 0; --k) {
#pragma unroll
          for (int m = 0; m < 3; ++m)
            if (lane + 64 * m < 169) { MB[pe[m]] = pm[m]; TB[pe[m]] = pt[m]; }
          {
            const int kn = max(k - 1, 0);
#pragma unroll
            for (int m = 0; m < 3; ++m) { pm[m] = Mg_[kn * 169 + pe[m]]; pt[m] = TAg_[k * 169 + pe[m]]; }
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
        PCLK(if (lane == 0) st.phase_clk[20] = clock64());
        double yprev = 0.0;
        for (int k = 0; k < F; ++k) {
#pragma unroll
          for (int m = 0; m < 3; ++m)
            if (lane + 64 * m < 169) { MB[pe[m]] = pm[m]; TB[pe[m]] = pt[m]; }
          {
            const int kn = min(k + 1, F - 1);
#pragma unroll
            for (int m = 0; m < 3; ++m) { pm[m] = Mg_[kn * 169 + pe[m]]; pt[m] = TAg_[kn * 169 + pe[m]]; }
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
      } else {
        // landmarks: y_l = (g_l - w_l^T yP) / (E_l + mu dhat_l^2), all 80 coupling entries of a landmark in flight at once
        const double *vP = lds + MW_VP;
        for (int l = lane; l < L; l += 64) {
          const double gl = lm_g[l], ei = lm_einv[l], d2 = lm_dh2[l], vl = gl / d2;
          double tl = 0.0, tq = 0.0;
          // (40 coupling entries of the landmark in flight at a time: the register budget of two waves per SIMD)
#pragma unroll
          for (int a0 = 0; a0 < 80; a0 += 40) {
            double wcol[40];
#pragma unroll
            for (int a = 0; a < 40; ++a) wcol[a] = wl[(size_t)(a0 + a) * L + l];
#pragma unroll
            for (int a = 0; a < 40; ++a)
              if (a0 + a < VILO_NPU) { tl += wcol[a] * y[a0 + a]; tq += wcol[a] * vP[a0 + a]; }   // y and v_P are zero on inactive dimensions
          }
          const double yl = (gl - tl) * ei;
          lm_y[l] = yl;
          part_gnn += d2 * yl * yl;
          part_gy += gl * yl;
          part_qx += vl * tq;   // cross term of q = v^T H v: 2 v_l w_l^T v_P
        }
        for (int cd = lane; cd < 80; cd += 64) {
          part_gnn += dh2[cd] * y[cd] * y[cd];
          part_gy += g[cd] * y[cd];
        }
      }
      PCLK(if (lane == 0) st.phase_clk[wave ? 7 : 21] = clock64());
      part_gnn = wave_sum(part_gnn); part_gy = wave_sum(part_gy); part_qx = wave_sum(part_qx);
      if (lane == 0) { red[MR_GNN + wave] = part_gnn; red[MR_GY + wave] = part_gy; if (wave == 1) red[MR_QX] = part_qx; }
      wg_barrier_global();   // (lm_y is read by both waves for the candidate)
      gnnorm2 = red[MR_GNN] + red[MR_GNN + 1];
      gy = red[MR_GY] + red[MR_GY + 1];
      qq = bimg[BI_SCAL + 0] + (q_lm + 2.0 * red[MR_QX]);
      PCLK(if (tid == 0) st.phase_clk[8] = clock64());
      if (!(isfinite(gnnorm2) && isfinite(gy))) {   // IsArrayValid(gauss_newton_step_) failed
        mu *= 10.0;
        if (tid == 0) { st.mu = mu; st.pad[0]++; }
        if (!(mu < 1.0)) {
          if (tid == 0) { st.lin_fail = 1; st.step_valid = 0; st.scale_ready = 1; }
          return;
        }
        wg_barrier();
        continue;
      }
      solved = true;
    }
    // keep the linearisation's vectors for the steps that reuse it after a rejected candidate
    double *cam_g = b.cam_g + (size_t)win * CD_N, *cam_dh2 = b.cam_dh2 + (size_t)win * CD_N, *cam_y = b.cam_y + (size_t)win * CD_N;
    if (wave == 1) {
      for (int cd = lane; cd < 80; cd += 64) { cam_g[cd] = g[cd]; cam_dh2[cd] = dh2[cd]; cam_y[cd] = y[cd]; }
    } else {
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int e = lane + 64 * m;
        if (e < 144) { cam_g[CD_B0 + e] = gBr[m]; cam_dh2[CD_B0 + e] = dBr[m]; cam_y[CD_B0 + e] = yBr[m]; }
      }
    }
    if (tid == 0) {
      st.gnorm2 = gnorm2; st.gnnorm2 = gnnorm2; st.gdotgn = -gy; st.q = qq; st.gmax = gmax;
      st.alpha = gnorm2 / qq;
      st.scale_ready = 1;
      st.lin_fail = 0;
    }
  } else {
    const double *cam_g = b.cam_g + (size_t)win * CD_N, *cam_dh2 = b.cam_dh2 + (size_t)win * CD_N, *cam_y = b.cam_y + (size_t)win * CD_N;
    if (wave == 1) {
      for (int cd = lane; cd < 80; cd += 64) { g[cd] = cam_g[cd]; dh2[cd] = cam_dh2[cd]; y[cd] = cam_y[cd]; }
    } else {
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int e = min(lane + 64 * m, 143);
        gBr[m] = cam_g[CD_B0 + e]; dBr[m] = cam_dh2[CD_B0 + e]; yBr[m] = cam_y[CD_B0 + e];
      }
    }
  }

  // ---- dogleg step for the current radius, candidate camera state ----
  if (tid == 0) {
    double ca = 0.0, cb = 0.0;
    int go = 0;
    // (st's scalars were written by this thread above: program order)
    if (st.radius <= sp.min_radius) { st.done = 1; st.termination = 1; st.step_valid = 0; }
    else { dogleg_scalars(st); ca = st.coef_a; cb = st.coef_b; go = st.step_valid; }
    red[MR_CA] = ca; red[MR_CB] = cb; red[MR_GO] = (double)go;
  }
  wg_barrier();
  const double ca = red[MR_CA], cb = red[MR_CB];
  const int go = (red[MR_GO] != 0.0) ? 1 : 0;
  const double *x = b.x + (size_t)win * XSTRIDE;
  double *xc = b.xc + (size_t)win * XSTRIDE;
  {
    // candidate inverse depths: lambda_c = lambda - a g_l / dhat_l^2 - b y_l (no valid step: the candidate is the current point)
    const double *lam = b.lam + wm.lm_off, *lmg = b.lm_gbuf[st.cur] + wm.lm_off, *lmd = b.lm_dh2 + wm.lm_off, *lmy = b.lm_y + wm.lm_off;
    double *lamc = b.lamc + wm.lm_off;
    for (int l = tid; l < L; l += 128) lamc[l] = go ? lam[l] - ca * lmg[l] / lmd[l] - cb * lmy[l] : lam[l];
  }
  if (!go) {
    for (int e = tid; e < XSTRIDE; e += 128) xc[e] = x[e];
    return;
  }
  double *del = lds + MW_T + MX_DEL;
  if (wave == 1) {
    for (int cd = lane; cd < 80; cd += 64) del[cd] = -ca * g[cd] / dh2[cd] - cb * y[cd];
  } else {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int e = lane + 64 * m;
      if (e < 143) del[CD_B0 + e] = -ca * gBr[m] / dBr[m] - cb * yBr[m];
    }
  }
  wg_barrier();
  if (wave == 1) {
    if (lane < 11) pose_plus(x + XO_POSE + 7 * lane, del + 6 * lane, xc + XO_POSE + 7 * lane);
    else if (lane < 13) pose_plus(x + XO_EX + 7 * (lane - 11), del + CD_EX0 + 6 * (lane - 11), xc + XO_EX + 7 * (lane - 11));
    else if (lane == 13) xc[XO_TD] = x[XO_TD] + del[CD_TD];
  } else {
    for (int e = lane; e < 143; e += 64) {
      const int k = e / 13, c = e - 13 * k;
      if (c < 9) xc[XO_SB + 9 * k + c] = x[XO_SB + 9 * k + c] + del[CD_B0 + e];
      else xc[XO_LB + 4 * k + (c - 9)] = x[XO_LB + 4 * k + (c - 9)] + del[CD_B0 + e];
    }
  }
  PCLK(if (tid == 0) st.phase_clk[9] = clock64());
}

int vilo_launch_mw_solver(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s) {
  const size_t lds_bytes = (size_t)MW_TOTAL * sizeof(double);
  hipLaunchKernelGGL(k_solve_mw, dim3(b.W), dim3(128), lds_bytes, s, b, sp);
  return VILO_OK;
}
