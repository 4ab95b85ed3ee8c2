// Four-wave solver for gfx950: k_solve_mw's algorithm (kernels_mw.hip; what ceres::Solve does per linearisation for
// Estimator::optimization(), estimator.cpp:1221-1236 — DENSE_SCHUR + traditional DOGLEG, Ceres 1.14 semantics) with the two serial parts of
// a window's solve cut in half again, for batches that leave most of the chip idle (one workgroup = one window per CU, up to 256 windows):
//
//   waves A1 / A2   the block-tridiagonal Cholesky of the speed / leg-bias part as a TWISTED factorisation: A1 eliminates frames F-1 .. m+1
//                   downwards, A2 frames 0 .. m-1 upwards, at the same time; the middle frame m = (F-1)/2 takes the Schur updates of both
//                   neighbours and is factorised last (by A1). The back-substitution sweeps run the same way: both halves forward, the
//                   middle, both halves backward. M_k = L_k^-1 and the off-diagonal factors stay in LDS (no round trip through L2).
//                     downwards:  T_A(k)  = L_k^-1 A_{k,k-1}      S_{k-1} = A_{k-1,k-1} - T_A(k)^T T_A(k)      T(k) = L_k^-1 (V_k - T_A(k+1)^T T(k+1))
//                     upwards:    T'_A(k) = L_k^-1 A_{k+1,k}^T    S_{k+1} = A_{k+1,k+1} - T'_A(k)^T T'_A(k)    T(k) = L_k^-1 (V_k - T'_A(k-1)^T T(k-1))
//                     middle:     S_m = A_mm - T_A(m+1)^T T_A(m+1) - T'_A(m-1)^T T'_A(m-1),   V_m loses both neighbours' terms
//   waves B1 / B2   the 80 x 80 pose system split by tiles (8 + 7 of the 15 lower 16 x 16 tiles): each runs the landmark Schur complement and
//                   the rank updates C -= T(k)^T T(k) on its own tiles (T(k) from both chain waves through LDS); B2's tiles then move to B1,
//                   which runs the blocked Cholesky and the backward solve as the two-wave form does; both share the landmark
//                   back-substitution.
// A different elimination order of the speed / leg-bias part and different partial sums: results agree with the other forms to rounding
// (tests run every form against the oracle at the same tolerances); batch-of-N == batch-of-1 bitwise within the form.
#include <type_traits>
#include "wave_common.hpp"

// LDS map (doubles)
#define Q_G 0          // [80]  gradient of the pose part
#define Q_DH2 80       // [80]  dogleg diagonal
#define Q_Y 160        // [80]  Gauss-Newton step of the pose part
#define Q_V 240        // [80]  reduced right-hand side
#define Q_VP 320       // [80]  v_P = D^-2 g
#define Q_DB 400       // [144] dogleg diagonal of the speed / leg-bias part
#define Q_GB 544       // [144] its gradient
#define Q_RED 688      // [64]  cross-wave sums and flags
#define Q_CH1 752      // [704] A1's chain scratch
#define Q_CH2 1456     // [704] A2's
#define Q_XSN 2160     // [176] T'_A(m-1)^T T'_A(m-1): A2's update of the middle frame's diagonal block
#define Q_XU 2336      // [16]  u of frame m - 1 (forward sweep hand-over)
#define Q_XY 2352      // [16]  y of frame m (backward sweep hand-over)
#define Q_M 2368       // [11][176] M_k = L_k^-1
#define Q_TA 4304      // [11][176] T_A(k) (frames above the middle), T'_A(k) (frames below)
#define Q_T1 6240      // [2][1280] A1's T(k) hand-over;  after the chain: Cholesky scratch (832) and panel slots (1024 at +1024)
#define Q_T2 8800      // [2][1280] A2's;                  after the chain: B2's seven tiles (1792), the step (224 at +2048)
#define Q_U 11360      // [144]
#define Q_YB 11504     // [144]
#define Q_SKIP 11648   // [40]
#define Q_TOTAL 11688
// chain scratch (per chain wave)
#define MC_LM 0
#define MC_TA0 176
#define MC_TA1 352
#define MC_SN 528
// Cholesky (Q_T1 region)
#define MX_D16 0
#define MX_LI16 272
#define MX_P16 560
#define MX_PANEL 1024
#define MX_DEL 2048   // (in Q_T2)
// reduction slots
#define QR_GN 0        // [4]
#define QR_GMAX 4      // [4]
#define QR_Q 8         // [4]
#define QR_FAIL 12     // [4] chain up, chain down, pose system
#define QR_GNN 16      // [4]
#define QR_GY 20       // [4]
#define QR_QX 24       // [4]
#define QR_CA 28
#define QR_CB 29
#define QR_GO 30

extern "C" size_t vilo_solve_mw4_lds_bytes() { return (size_t)Q_TOTAL * sizeof(double); }

__device__ __forceinline__ void q_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void q_barrier_global() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// (profiling build: cycles a wave spends at the workgroup barriers, phase_clk[20 + wave]; its total, phase_clk[16 + wave])
#define Q_BARRIER() do { PCLK(qw0 = clock64()); q_barrier(); PCLK(qwait += clock64() - qw0); } while (0)
#define Q_BARRIER_GLOBAL() do { PCLK(qw0 = clock64()); q_barrier_global(); PCLK(qwait += clock64() - qw0); } while (0)
// tiles of the pose system owned by matrix wave 0 / 1 (bit t of the mask = tile t of c_tI / c_tJ)
//   B1: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) (3,0) (3,1)   = tiles 0 .. 7;   B2: (3,2) (3,3) (4,0) .. (4,4) = tiles 8 .. 14
#define B1_MASK 0x00ff
#define B2_MASK 0x7f00

__global__ void __launch_bounds__(256) k_solve_mw4(BatchDev b, SolveParams sp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int win = blockIdx.x;
  SolverState &st = b.st[win];
  if (st.done) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0: A1 (chain down + middle), 1: A2 (chain up), 2: B1, 3: B2
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int tid = threadIdx.x;
  long long qwait = 0, qw0 = 0;   // (profiling build only)
  const long long q_start = pclk64();
#define QSTAMP(w_, i_) PCLK(if (wave == (w_) && lane == 0) st.phase_clk[i_] = clock64() - q_start)
  const WinMeta wm = b.win[win];
  const int F = wm.n_frames, L = wm.L, kb = wm.pad, cmask = wm.const_mask;
  const int mid = (F - 1) >> 1, nL = mid, nU = F - 1 - mid, nS = nU;   // frames below / above the middle; nU >= nL, nU >= 1
  double *g = lds + Q_G, *dh2 = lds + Q_DH2, *y = lds + Q_Y, *v = lds + Q_V, *red = lds + Q_RED;
  double *DB = lds + Q_DB, *GB = lds + Q_GB, *U = lds + Q_U, *YB = lds + Q_YB;
  // the dogleg's inputs of a fresh linearisation stay in lane 0's registers (tid 0); the radius is the bookkeeping's, read up front
  struct DoglegIn { double radius, mu, gnorm2, gnnorm2, gdotgn, q, alpha, coef_a, coef_b, dogleg_step_norm, model_cost_change; int step_valid; } fresh;
  bool have_fresh = false;
  const double radius0 = st.radius;

  if (st.need_lin) {
    const double *bimg = b.Bimg + (size_t)win * BI_N;
    const double *gin = b.cam_gin + (size_t)win * CD_N;
    const double *wl = b.lm_w + 80 * (size_t)wm.lm_off;
    double *lm_E = b.lm_E + wm.lm_off, *lm_g = b.lm_gbuf[st.cur] + wm.lm_off, *lm_dh2 = b.lm_dh2 + wm.lm_off, *lm_scale = b.lm_scale + wm.lm_off,
           *lm_einv = b.lm_einv + wm.lm_off, *lm_y = b.lm_y + wm.lm_off;
    const bool first_scale = !st.scale_ready;
    double mu = st.mu;

    // ---- camera-side vectors come scaled from k_assemble; the landmarks are scaled here (landmark tid + 256 n) ----
    double part_gn = 0.0, part_gmax = 0.0, part_q = 0.0;
    if (wave == 2) {
      for (int cd = lane; cd < 80; cd += 64) { g[cd] = gin[cd]; dh2[cd] = bimg[BI_DH2 + cd]; lds[Q_VP + cd] = cd_active(cd, F, cmask) ? bimg[BI_V + cd] : 0.0; }
    } else if (wave == 0) {
      for (int e = lane; e < 144; e += 64) { DB[e] = bimg[BI_DH2 + CD_B0 + min(e, 143)]; GB[e] = gin[CD_B0 + min(e, 143)]; }
    }
    for (int l = tid; l < L; l += 256) {
      const double E = lm_E[l], gl = lm_g[l];
      double sc;
      if (first_scale) { sc = sp.jacobi_scaling ? 1.0 / (1.0 + sqrt(E)) : 1.0; lm_scale[l] = sc; }
      else sc = lm_scale[l];
      const double d2 = fmin(fmax(sc * sc * E, sp.min_lm_diagonal), sp.max_lm_diagonal) / (sc * sc);
      lm_dh2[l] = d2;
      lm_einv[l] = 1.0 / (E + mu * d2);   // (of the first factorisation: a retry at a larger mu forms its own below)
      const double vl = gl / d2;
      part_q += E * vl * vl;   // (the cross term 2 vl w_l^T v_P comes from the back-substitution)
      part_gn += gl * vl;
      part_gmax = fmax(part_gmax, fabs(gl));
    }
    part_gn = wave_sum(part_gn); part_gmax = wave_max(part_gmax); part_q = wave_sum(part_q);
    if (lane == 0) { red[QR_GN + wave] = part_gn; red[QR_GMAX + wave] = part_gmax; red[QR_Q + wave] = part_q; }
    Q_BARRIER_GLOBAL();
    const double gnorm2 = bimg[BI_SCAL + 1] + (((red[QR_GN] + red[QR_GN + 1]) + red[QR_GN + 2]) + red[QR_GN + 3]);
    const double gmax = fmax(bimg[BI_SCAL + 2], fmax(fmax(red[QR_GMAX], red[QR_GMAX + 1]), fmax(red[QR_GMAX + 2], red[QR_GMAX + 3])));
    const double q_lm = ((red[QR_Q] + red[QR_Q + 1]) + red[QR_Q + 2]) + red[QR_Q + 3];
    if (!sp.fixed_iterations && gmax <= sp.gradient_tolerance) {
      if (tid == 0) { st.gmax = gmax; st.done = 1; st.termination = 1; st.step_valid = 0; }
      return;
    }

    QSTAMP(2, 0);   // prologue: scaling of the landmarks
    bool solved = false, retry = false;
    double qq = 0.0, gnnorm2 = 0.0, gy = 0.0;
    const int lane_outer = lane;
    while (!solved) {
      int lane = lane_outer;
      asm volatile("" : "+v"(lane));
      const int lr = lane & 15, lk = lane >> 4;
      if (retry) {
        for (int l = tid; l < L; l += 256) lm_einv[l] = 1.0 / (lm_E[l] + mu * lm_dh2[l]);
        Q_BARRIER_GLOBAL();
      }
      retry = true;
      const int nks = (L + 3) >> 2, NT = (nks + 3) >> 2, NH = (nks + 1) >> 1;   // k-steps of 4 landmarks, trips of 4, half trips of 2
      int fail = 0;

      if (wave < 2) {
        // =============================== waves A1 / A2: the twisted chain ===============================
        const bool up = (wave == 1);   // A2 walks the frames upwards from 0
        int fX[5], oX[5];
#pragma unroll
        for (int X = 0; X < 5; ++X) { const int col = 16 * X + lr; fX[X] = col < 66 ? col / 6 : 99; oX[X] = col < 66 ? col - 6 * fX[X] : 0; }
        const int grp = lk, c = lr;
        const int row = c < 13 ? c : 0;
        double *scr = lds + (up ? Q_CH2 : Q_CH1);
        double *LM = scr + MC_LM, *SN = scr + MC_SN;
        double *TAcur = scr + MC_TA0, *TAprev = scr + MC_TA1;
        double *Tbase = lds + (up ? Q_T2 : Q_T1);
        mfma_d4 T[5];
#pragma unroll
        for (int X = 0; X < 5; ++X) T[X] = mfma_d4{0.0, 0.0, 0.0, 0.0};
        // one frame of a chain. nb: the neighbour this frame hands its Schur update to (-1: none: the middle frame); prevk: the frame
        // eliminated before this one in the same direction (-1: first)
        auto frame = [&](int k, int nb, int prevk, bool middle, int step) {
          const int x_lo = (up || middle || k <= kb) ? 0 : max(0, (6 * (k - 1)) >> 4);   // (upwards and in the middle T(k) is dense)
          mfma_d4 V[5];
          double a[13], l[13], rhs[13], adn[4];
          // [B_k | g_k] in accumulator order: row lk + 4 r, column 16 X + lr; column 79 carries the gradient
#pragma unroll
          for (int X = 0; X < 5; ++X) {
            const int df = fX[X] - k + 1;
            const bool on = df >= 0 && df <= 2;
            const double *src = bimg + BI_BS + (k * 16 + lk) * 18 + 6 * min(max(df, 0), 2) + oX[X];
#pragma unroll
            for (int r = 0; r < 4; ++r) V[X][r] = on ? src[72 * r] : 0.0;
          }
          if (k == kb) {
#pragma unroll
            for (int X = 0; X < 5; ++X)
#pragma unroll
              for (int r = 0; r < 4; ++r) V[X][r] += bimg[BI_BP + (lk + 4 * r) * 80 + 16 * X + lr];
          }
          // right-hand sides of the off-diagonal factor: column `row` of A_{k,k-1} (downwards), of A_{k+1,k}^T (upwards)
#pragma unroll
          for (int i = 0; i < 13; ++i) {
            double rv = 0.0;
            if (nb >= 0) rv = up ? bimg[BI_AOT + (k * 13 + i) * 13 + row] : bimg[BI_AOT + ((k - 1) * 13 + row) * 13 + i];
            rhs[i] = rv;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) adn[r] = (nb >= 0 && lr < 13 && lk + 4 * r < 13) ? bimg[BI_AD + nb * 169 + (lk + 4 * r) * 13 + lr] : 0.0;   // A_{nb,nb}, accumulator order
          if (lr == 15) {
#pragma unroll
            for (int r = 0; r < 4; ++r) V[4][r] = (lk + 4 * r < 13) ? GB[13 * k + lk + 4 * r] : 0.0;
          }
          // S_k (lane = row): the first frame of a direction straight from A_kk, later ones from the update left by the previous step
          if (prevk < 0) {
#pragma unroll
            for (int j = 0; j < 13; ++j) a[j] = bimg[BI_AD + (k * 13 + row) * 13 + j];
          } else {
#pragma unroll
            for (int j = 0; j < 13; ++j) a[j] = SN[row * 13 + j];
          }
          if (middle && nL > 0) {
            const double *XS = lds + Q_XSN;
#pragma unroll
            for (int j = 0; j < 13; ++j) a[j] -= XS[row * 13 + j];
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
            __builtin_amdgcn_sched_barrier(0);
          }
          double *Mk = lds + Q_M + 176 * k, *TAk = lds + Q_TA + 176 * k;
          if (c < 13 && grp < 2) {
            if (grp == 0) {
#pragma unroll
              for (int q = 0; q < 13; ++q) { TAcur[q * 13 + c] = cl[q]; TAk[q * 13 + c] = cl[q]; }
            } else {
#pragma unroll
              for (int q = 0; q < 13; ++q) { LM[q * 13 + c] = cl[q]; Mk[q * 13 + c] = cl[q]; }
            }
          }
          lds_fence();
          // the neighbour's diagonal block loses T_A^T T_A: one 16 x 16 tile on the matrix cores. The frame next to the middle leaves A2's
          // share where A1 finds it (A1's own share goes into its SN with A_mm)
          if (nb >= 0) {
            mfma_d4 sn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const int q = 4 * kk + lk;
              const double ta = ((lr < 13) && (q < 13)) ? TAcur[min(q, 12) * 13 + min(lr, 12)] : 0.0;
              sn = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, ta, sn, 0, 0, 0);
            }
            const bool to_mid = up && nb == mid;
            double *dst = to_mid ? lds + Q_XSN : SN;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (lr < 13 && lk + 4 * r < 13) dst[(lk + 4 * r) * 13 + lr] = to_mid ? sn[r] : adn[r] - sn[r];
          }
          // V -= T_A(prev)^T T(prev);  T(k) = M_k V
          double at[4], am[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int q = 4 * kk + lk;
            const bool in = (lr < 13) && (q < 13);
            const double ta = TAprev[min(q, 12) * 13 + min(lr, 12)], m = LM[min(lr, 12) * 13 + min(q, 12)];
            at[kk] = (in && prevk >= 0) ? -ta : 0.0;
            am[kk] = in ? m : 0.0;
          }
          if (prevk >= 0) {
#pragma unroll
            for (int X = 0; X < 5; ++X)
              if (X >= x_lo) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) V[X] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[kk], T[X][kk], V[X], 0, 0, 0);
              }
          }
          if (middle && nL > 0) {
            // the lower neighbour's term: T'_A(m-1) from the factor store, T(m-1) from A2's hand-over buffer (its last frame)
            const double *TAl = lds + Q_TA + 176 * (mid - 1);
            const double *Tl = lds + Q_T2 + 1280 * ((nL - 1) & 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              const int q = 4 * kk + lk;
              const double ta = ((lr < 13) && (q < 13)) ? -TAl[min(q, 12) * 13 + min(lr, 12)] : 0.0;
#pragma unroll
              for (int X = 0; X < 5; ++X) V[X] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, Tl[(X * 4 + kk) * 64 + lane], V[X], 0, 0, 0);
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
          // hand T(k) over (tiles left of x_lo are zero: written as such, the consumers need no per-frame sparsity table)
          double *Tb = Tbase + 1280 * (step & 1);
#pragma unroll
          for (int X = 0; X < 5; ++X)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) Tb[(X * 4 + kk) * 64 + lane] = (X >= x_lo) ? T[X][kk] : 0.0;
          double *sw = TAcur; TAcur = TAprev; TAprev = sw;
        };
        Q_BARRIER();   // (the matrix waves' skip table: every wave meets the same barriers)
        for (int i = 0; i <= nS + 1; ++i) {
          if (!up) {
            if (i < nU) { const int k = F - 1 - i; frame(k, k - 1, i == 0 ? -1 : k + 1, false, i); }
            else if (i == nS) frame(mid, -1, mid + 1, true, i);
          } else if (i < nL) {
            frame(i, i + 1, i == 0 ? -1 : i - 1, false, i);
          }
          if (i == nS + 1 && lane == 0) red[QR_FAIL + wave] = (double)fail;
          Q_BARRIER();
        }
        QSTAMP(0, 8);   // A1: chain done (own work + step barriers)
        Q_BARRIER();   // B2's tiles and the reduced right-hand side
        Q_BARRIER();   // y_P
      } else {
        // =============================== waves B1 / B2: the pose system ===============================
        const int bw = wave - 2;
        const double *Cimg = b.Cimg + (size_t)win * CIMG_N;
        mfma_d4 acc[15];
#pragma unroll
        for (int t = 0; t < 15; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
        auto run = [&](auto mask_c) {
          constexpr int MASK = decltype(mask_c)::value;
          constexpr bool MAIN_RHS = (MASK == B2_MASK);   // B2 holds operands of all five tile rows: it forms the reduced right-hand side
#pragma unroll
          for (int t = 0; t < 15; ++t)
            if ((MASK >> t) & 1) {
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[t][r] = Cimg[(t * 4 + r) * 64 + lane];
            }
          double yacc[5], yr[5];
#pragma unroll
          for (int X = 0; X < 5; ++X) { yacc[X] = 0.0; yr[X] = 0.0; }
          int *skip_tab = (int *)(lds + Q_SKIP);
          if (bw == 0 && L > 0) {
            const unsigned char *lms = b.lm_s + wm.lm_off;
            for (int tr = lane; tr < NT + 2; tr += 64) skip_tab[tr] = (6 * (int)lms[min(16 * tr, L - 1)] >= 16) ? 1 : 0;
          }
          Q_BARRIER();   // (the skip table is B1's; both matrix waves read it — the chain waves meet this barrier too)
          constexpr int XMAX = MAIN_RHS ? 5 : 4;   // operand blocks the wave's tiles touch
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
              for (int X = 1; X < XMAX; ++X) opb[bsel][u][X] = wl[(size_t)(16 * X + lr) * L + lc];
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
                if (((MASK >> t) & 1) && c_tJ[t] >= XL)
                  acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-(opb[bsel][u][c_tI[t]] * ei), opb[bsel][u][c_tJ[t]], acc[t], 0, 0, 0);
              if (MAIN_RHS) {
#pragma unroll
                for (int X = XL; X < 5; ++X) yacc[X] += opb[bsel][u][X] * ge;
              }
            }
          };
          auto half = [&](int h, auto bs) {
            const int skip0 = __builtin_amdgcn_readfirstlane(skip_tab[min(h >> 1, NT)]);
            if (skip0) dohalf(bs, std::integral_constant<int, 1>{});
            else dohalf(bs, std::integral_constant<int, 0>{});
          };
          // C -= T^T T for one delivered frame (all five tile columns: the chain writes zeros where T(k) has none)
          auto rank_update = [&](const double *Tb) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              double Tk[5];
#pragma unroll
              for (int X = 0; X < 5; ++X) Tk[X] = Tb[(X * 4 + kk) * 64 + lane];
              const double tg = __shfl(Tk[4], (lane & 48) | 15, 64);   // t_g(k) (column 79): the pose system must not see it
              if (lr == 15) Tk[4] = 0.0;
#pragma unroll
              for (int t = 0; t < 15; ++t)
                if ((MASK >> t) & 1) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Tk[c_tI[t]], Tk[c_tJ[t]], acc[t], 0, 0, 0);
              if (MAIN_RHS) {
#pragma unroll
                for (int X = 0; X < 5; ++X) yr[X] += Tk[X] * tg;
              }
            }
          };
          if (NH > 0) ldhalf(0, std::integral_constant<int, 0>{});
          int h = 0;
          for (int i = 0; i <= nS + 1; ++i) {
            const int h_end = ((i + 1) * NH) / (nS + 2);
            for (; h < h_end; ++h) {
              if (h & 1) {
                if (h + 1 < NH) ldhalf(h + 1, std::integral_constant<int, 0>{});
                half(h, std::integral_constant<int, 1>{});
              } else {
                if (h + 1 < NH) ldhalf(h + 1, std::integral_constant<int, 1>{});
                half(h, std::integral_constant<int, 0>{});
              }
            }
            if (i >= 1) {
              // what the chain waves finished in step i - 1: A1 a frame above the middle or (step nS) the middle, A2 a frame below
              if (i - 1 < nU || i - 1 == nS) rank_update(lds + Q_T1 + 1280 * ((i - 1) & 1));
              if (i - 1 < nL) rank_update(lds + Q_T2 + 1280 * ((i - 1) & 1));
            }
            Q_BARRIER();
          }
          if (MAIN_RHS) {
            // reduced right-hand side: g_P - sum_k T_B^T t_g - sum_l w_l g_l / (E_l + mu dhat_l^2)
#pragma unroll
            for (int X = 0; X < 5; ++X) {
              double s = yr[X] + yacc[X];
              s += __shfl_xor(s, 16, 64);
              s += __shfl_xor(s, 32, 64);
              if (lk == 0) v[16 * X + lr] = g[16 * X + lr] - s;
            }
            // B2's tiles move to B1 (accumulator order, one coalesced LDS store per register)
            double *Gt = lds + Q_T2;
#pragma unroll
            for (int t = 8; t < 15; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) Gt[((t - 8) * 4 + r) * 64 + lane] = acc[t][r];
          }
        };
        if (bw == 0) run(std::integral_constant<int, B1_MASK>{});
        else run(std::integral_constant<int, B2_MASK>{});
        QSTAMP(2, 1);   // Schur complement + rank updates (own work done)
        Q_BARRIER();   // B2's tiles and the reduced right-hand side are there
        QSTAMP(2, 2);
        if (bw == 0) {
          fail = (red[QR_FAIL] != 0.0 || red[QR_FAIL + 1] != 0.0) ? 1 : 0;
          if (!fail) {
            // ---- B1: the whole pose system in registers, regularised (diag += mu dhat^2); blocked Cholesky (diagonal tile in registers +
            //      v_readlane, panel and trailing update on the matrix cores, the right-hand side riding along as a sixth block row);
            //      backward solve with the factor's accumulator registers as operands: as in k_solve_wave / k_solve_mw ----
            const double *Gt = lds + Q_T2;
#pragma unroll
            for (int t = 8; t < 15; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[t][r] = Gt[((t - 8) * 4 + r) * 64 + lane];
#pragma unroll
            for (int I = 0; I < 5; ++I)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (lk + 4 * r == lr) acc[tile_index(I, I)][r] += mu * dh2[16 * I + lr];
            double *scr = lds + Q_T1;
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
                for (int r = 0; r < 4; ++r) lds[pswz_at(Q_T1 + MX_PANEL, I - j - 1, lk + 4 * r, lr)] = nacc[r];
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
                for (int kk = 0; kk < 4; ++kk) pa[I][kk] = lds[pswz_at(Q_T1 + MX_PANEL, I - j - 1, lr, 4 * kk + lk)];
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
          QSTAMP(2, 3);   // blocked Cholesky + forward solve
          if (!fail) {
            // L^T yP = y, blockwise on the matrix cores: x_j = L_jj^-T (y_j - sum_{i>j} L_ij^T x_i)
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
          if (lane == 0) red[QR_FAIL + 2] = (double)fail;
          QSTAMP(2, 4);   // backward solve
        }
        Q_BARRIER();   // y_P (or the failure flags)
      }
      fail = 0;
      // (wave-uniform flags from LDS: chain up / down, pose system)
      if (red[QR_FAIL] != 0.0 || red[QR_FAIL + 1] != 0.0 || red[QR_FAIL + 2] != 0.0) fail = 1;
      if (fail) {
        // DoglegStrategy::ComputeGaussNewtonStep: mu *= 10 and retry while mu < max_mu (1.0)
        mu *= 10.0;
        if (tid == 0) { st.mu = mu; st.pad[0]++; }   // (pad[0]: factorisation retries of this solve, read by the tests)
        if (!(mu < 1.0)) {
          if (tid == 0) { st.lin_fail = 1; st.step_valid = 0; st.gnorm2 = gnorm2; st.q = 0.0; st.gmax = gmax; st.scale_ready = 1; }
          return;
        }
        Q_BARRIER();   // (every lane has read the flags before the next trip rewrites them)
        continue;
      }

      // ---- back-substitution: A1 / A2 the speed / leg-bias part (twisted sweeps), B1 / B2 the landmarks ----
      double part_gnn = 0.0, part_gy = 0.0, part_qx = 0.0;
      if (wave < 2) {
        const bool up = (wave == 1);
        const int k_lo = up ? 0 : mid, k_hi = up ? mid - 1 : F - 1;   // this wave's frames (A1 owns the middle)
        // c = g_B - B yP for this wave's frames: the IMU part of B_k spans poses k-1 .. k+1, the prior part frame kb only
        {
          double bsv[3][18];
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const int e = min(lane + 64 * m, 142), k = e / 13;
#pragma unroll
            for (int s2 = 0; s2 < 18; ++s2) bsv[m][s2] = bimg[BI_BS + (16 * k + (e - 13 * k)) * 18 + s2];
          }
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const int e = lane + 64 * m, k = min(e, 142) / 13;
            double sacc = GB[min(e, 143)];
#pragma unroll
            for (int s2 = 0; s2 < 18; ++s2) sacc -= bsv[m][s2] * y[min(max(6 * (k - 1) + s2, 0), 79)];
            if (e < 143 && k >= k_lo && k <= k_hi) U[e] = sacc;
          }
        }
        lds_fence();
        if (kb >= k_lo && kb <= k_hi) {
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
        const double *Mall = lds + Q_M, *TAall = lds + Q_TA;
        // forward sweeps, both halves at once:   down  u_k = M_k (c_k - T_A(k+1)^T u_{k+1})   k = F-1 .. m+1
        //                                        up    u_k = M_k (c_k - T'_A(k-1)^T u_{k-1})  k = 0 .. m-1
        double uprev = 0.0;
        if (!up) {
          for (int k = F - 1; k > mid; --k) {
            double s2 = U[13 * k + row];
            if (k < F - 1) {
              const double *TB = TAall + 176 * (k + 1);
#pragma unroll
              for (int q = 0; q < 13; ++q) s2 -= TB[q * 13 + row] * readlane_d(uprev, q);
            }
            const double *MB = Mall + 176 * k;
            double u = 0.0;
#pragma unroll
            for (int q = 0; q < 13; ++q) u += MB[row * 13 + q] * readlane_d(s2, q);
            if (lane < 13) U[13 * k + lane] = u;
            uprev = u;
          }
        } else {
          for (int k = 0; k < mid; ++k) {
            double s2 = U[13 * k + row];
            if (k > 0) {
              const double *TB = TAall + 176 * (k - 1);
#pragma unroll
              for (int q = 0; q < 13; ++q) s2 -= TB[q * 13 + row] * readlane_d(uprev, q);
            }
            const double *MB = Mall + 176 * k;
            double u = 0.0;
#pragma unroll
            for (int q = 0; q < 13; ++q) u += MB[row * 13 + q] * readlane_d(s2, q);
            if (lane < 13) U[13 * k + lane] = u;
            uprev = u;
          }
          if (lane < 13) lds[Q_XU + lane] = uprev;   // u of frame m - 1 for the middle
        }
        Q_BARRIER();
        // the middle frame (A1): u_m = M_m (c_m - T_A(m+1)^T u_{m+1} - T'_A(m-1)^T u_{m-1}),  y_m = M_m^T u_m
        double ymid = 0.0;
        if (!up) {
          double s2 = U[13 * mid + row];
          if (nU > 0) {
            const double *TB = TAall + 176 * (mid + 1);
#pragma unroll
            for (int q = 0; q < 13; ++q) s2 -= TB[q * 13 + row] * readlane_d(uprev, q);
          }
          if (nL > 0) {
            const double *TB = TAall + 176 * (mid - 1);
            const double ul = lds[Q_XU + row];
#pragma unroll
            for (int q = 0; q < 13; ++q) s2 -= TB[q * 13 + row] * readlane_d(ul, q);
          }
          const double *MB = Mall + 176 * mid;
          double u = 0.0;
#pragma unroll
          for (int q = 0; q < 13; ++q) u += MB[row * 13 + q] * readlane_d(s2, q);
          double yk = 0.0;
#pragma unroll
          for (int q = 0; q < 13; ++q) yk += MB[q * 13 + row] * readlane_d(u, q);
          if (!cd_active(CD_B0 + 13 * mid + row, F, cmask)) yk = 0.0;
          if (lane < 13) { YB[13 * mid + lane] = yk; lds[Q_XY + lane] = yk; }
          ymid = yk;
        }
        Q_BARRIER();
        // backward sweeps, both halves at once:  down  y_k = M_k^T (u_k - T_A(k) y_{k-1})    k = m+1 .. F-1
        //                                        up    y_k = M_k^T (u_k - T'_A(k) y_{k+1})   k = m-1 .. 0
        double yprev = up ? lds[Q_XY + row] : ymid;
        if (!up) {
          for (int k = mid + 1; k < F; ++k) {
            const double *TB = TAall + 176 * k, *MB = Mall + 176 * k;
            double s2 = U[13 * k + row];
#pragma unroll
            for (int q = 0; q < 13; ++q) s2 -= TB[row * 13 + q] * readlane_d(yprev, q);
            double yk = 0.0;
#pragma unroll
            for (int q = 0; q < 13; ++q) yk += MB[q * 13 + row] * readlane_d(s2, q);
            if (!cd_active(CD_B0 + 13 * k + row, F, cmask)) yk = 0.0;
            if (lane < 13) YB[13 * k + lane] = yk;
            yprev = yk;
          }
        } else {
          for (int k = mid - 1; k >= 0; --k) {
            const double *TB = TAall + 176 * k, *MB = Mall + 176 * k;
            double s2 = U[13 * k + row];
#pragma unroll
            for (int q = 0; q < 13; ++q) s2 -= TB[row * 13 + q] * readlane_d(yprev, q);
            double yk = 0.0;
#pragma unroll
            for (int q = 0; q < 13; ++q) yk += MB[q * 13 + row] * readlane_d(s2, q);
            if (!cd_active(CD_B0 + 13 * k + row, F, cmask)) yk = 0.0;
            if (lane < 13) YB[13 * k + lane] = yk;
            yprev = yk;
          }
        }
        lds_fence();
        // norms of this wave's frames
        for (int e = lane; e < 143; e += 64) {
          const int k = e / 13;
          if (k >= k_lo && k <= k_hi && k < F) {
            const double yb = YB[e];
            part_gnn += DB[e] * yb * yb;   // (y is zero on inactive dimensions)
            part_gy += GB[e] * yb;
          }
        }
      } else {
        // landmarks: y_l = (g_l - w_l^T yP) / (E_l + mu dhat_l^2); landmark trips alternate between the two matrix waves
        const int bw = wave - 2;
        const double *vP = lds + Q_VP;
        for (int l = lane + 64 * bw; l < L; l += 128) {
          const double gl = lm_g[l], ei = lm_einv[l], d2 = lm_dh2[l], vl = gl / d2;
          double tl = 0.0, tq = 0.0;
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
        if (bw == 0) {
          for (int cd = lane; cd < 80; cd += 64) {
            part_gnn += dh2[cd] * y[cd] * y[cd];
            part_gy += g[cd] * y[cd];
          }
        }
        Q_BARRIER();   // (the chain waves' hand-over points)
        Q_BARRIER();
      }
      QSTAMP(2, 5);   // landmark back-substitution (+ waiting for the sweeps' hand-over barriers)
      QSTAMP(0, 9);   // A1: sweeps done
      part_gnn = wave_sum(part_gnn); part_gy = wave_sum(part_gy); part_qx = wave_sum(part_qx);
      if (lane == 0) { red[QR_GNN + wave] = part_gnn; red[QR_GY + wave] = part_gy; red[QR_QX + wave] = part_qx; }
      Q_BARRIER_GLOBAL();   // (lm_y is read by every wave for the candidate)
      gnnorm2 = ((red[QR_GNN] + red[QR_GNN + 1]) + red[QR_GNN + 2]) + red[QR_GNN + 3];
      gy = ((red[QR_GY] + red[QR_GY + 1]) + red[QR_GY + 2]) + red[QR_GY + 3];
      qq = bimg[BI_SCAL + 0] + (q_lm + 2.0 * (red[QR_QX + 2] + red[QR_QX + 3]));
      if (!(isfinite(gnnorm2) && isfinite(gy))) {   // IsArrayValid(gauss_newton_step_) failed
        mu *= 10.0;
        if (tid == 0) { st.mu = mu; st.pad[0]++; }
        if (!(mu < 1.0)) {
          if (tid == 0) { st.lin_fail = 1; st.step_valid = 0; st.scale_ready = 1; }
          return;
        }
        Q_BARRIER();
        continue;
      }
      solved = true;
    }
    // keep the linearisation's vectors for the steps that reuse it after a rejected candidate
    double *cam_g = b.cam_g + (size_t)win * CD_N, *cam_dh2 = b.cam_dh2 + (size_t)win * CD_N, *cam_y = b.cam_y + (size_t)win * CD_N;
    if (wave == 2) {
      for (int cd = lane; cd < 80; cd += 64) { cam_g[cd] = g[cd]; cam_dh2[cd] = dh2[cd]; cam_y[cd] = y[cd]; }
    } else if (wave == 0) {
      for (int e = lane; e < 144; e += 64) { cam_g[CD_B0 + e] = GB[e]; cam_dh2[CD_B0 + e] = DB[e]; cam_y[CD_B0 + e] = (e < 13 * F) ? YB[e] : 0.0; }
    }
    if (tid == 0) {
      st.gnorm2 = gnorm2; st.gnnorm2 = gnnorm2; st.gdotgn = -gy; st.q = qq; st.gmax = gmax;
      st.alpha = gnorm2 / qq;
      st.scale_ready = 1;
      st.lin_fail = 0;
      // (the dogleg below works on a register copy of these: no round trip through what was just stored)
      fresh.gnorm2 = gnorm2; fresh.gnnorm2 = gnnorm2; fresh.gdotgn = -gy; fresh.q = qq; fresh.alpha = gnorm2 / qq; fresh.mu = mu;
      have_fresh = true;
    }
  } else {
    const double *cam_g = b.cam_g + (size_t)win * CD_N, *cam_dh2 = b.cam_dh2 + (size_t)win * CD_N, *cam_y = b.cam_y + (size_t)win * CD_N;
    if (wave == 2) {
      for (int cd = lane; cd < 80; cd += 64) { g[cd] = cam_g[cd]; dh2[cd] = cam_dh2[cd]; y[cd] = cam_y[cd]; }
    } else if (wave == 0) {
      for (int e = lane; e < 144; e += 64) { GB[e] = cam_g[CD_B0 + min(e, 143)]; DB[e] = cam_dh2[CD_B0 + min(e, 143)]; YB[e] = cam_y[CD_B0 + min(e, 143)]; }
    }
  }

  QSTAMP(2, 6);   // norms, vectors kept
  // ---- dogleg step for the current radius, candidate camera state ----
  if (tid == 0) {
    double ca = 0.0, cb = 0.0;
    int go = 0;
    const double radius = radius0;   // (read when the kernel started)
    if (radius <= sp.min_radius) { st.done = 1; st.termination = 1; st.step_valid = 0; }
    else if (have_fresh) {
      fresh.radius = radius;
      dogleg_scalars(fresh);
      st.coef_a = fresh.coef_a; st.coef_b = fresh.coef_b; st.dogleg_step_norm = fresh.dogleg_step_norm; st.model_cost_change = fresh.model_cost_change;
      st.step_valid = fresh.step_valid;
      ca = fresh.coef_a; cb = fresh.coef_b; go = fresh.step_valid;
    }
    else { dogleg_scalars(st); ca = st.coef_a; cb = st.coef_b; go = st.step_valid; }
    red[QR_CA] = ca; red[QR_CB] = cb; red[QR_GO] = (double)go;
  }
  Q_BARRIER();
  const double ca = red[QR_CA], cb = red[QR_CB];
  const int go = (red[QR_GO] != 0.0) ? 1 : 0;
  const double *x = b.x + (size_t)win * XSTRIDE;
  double *xc = b.xc + (size_t)win * XSTRIDE;
  {
    const double *lam = b.lam + wm.lm_off, *lmg = b.lm_gbuf[st.cur] + wm.lm_off, *lmd = b.lm_dh2 + wm.lm_off, *lmy = b.lm_y + wm.lm_off;
    double *lamc = b.lamc + wm.lm_off;
    for (int l = tid; l < L; l += 256) lamc[l] = go ? lam[l] - ca * lmg[l] / lmd[l] - cb * lmy[l] : lam[l];
  }
  if (!go) {
    for (int e = tid; e < XSTRIDE; e += 256) xc[e] = x[e];
    return;
  }
  double *del = lds + Q_T2 + MX_DEL;
  if (wave == 2) {
    for (int cd = lane; cd < 80; cd += 64) del[cd] = -ca * g[cd] / dh2[cd] - cb * y[cd];
  } else if (wave == 0) {
    for (int e = lane; e < 143; e += 64) del[CD_B0 + e] = -ca * GB[e] / DB[e] - cb * YB[e];
  }
  Q_BARRIER();
  if (wave == 2) {
    if (lane < 11) pose_plus(x + XO_POSE + 7 * lane, del + 6 * lane, xc + XO_POSE + 7 * lane);
    else if (lane < 13) pose_plus(x + XO_EX + 7 * (lane - 11), del + CD_EX0 + 6 * (lane - 11), xc + XO_EX + 7 * (lane - 11));
    else if (lane == 13) xc[XO_TD] = x[XO_TD] + del[CD_TD];
  } else if (wave == 0) {
    for (int e = lane; e < 143; e += 64) {
      const int k = e / 13, c = e - 13 * k;
      if (c < 9) xc[XO_SB + 9 * k + c] = x[XO_SB + 9 * k + c] + del[CD_B0 + e];
      else xc[XO_LB + 4 * k + (c - 9)] = x[XO_LB + 4 * k + (c - 9)] + del[CD_B0 + e];
    }
  }
  QSTAMP(2, 7);   // dogleg + candidate
  PCLK(if (lane == 0) { st.phase_clk[16 + wave] = clock64() - q_start; st.phase_clk[20 + wave] = qwait; });
}

int vilo_launch_mw4_solver(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s) {
  const size_t lds_bytes = (size_t)Q_TOTAL * sizeof(double);
  if (!ctx->mw4_attr_set) {   // (per context = per device, like the other dynamic-LDS opt-ins)
    VILO_HIP(hipFuncSetAttribute((const void *)k_solve_mw4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    ctx->mw4_attr_set = true;
  }
  hipLaunchKernelGGL(k_solve_mw4, dim3(b.W), dim3(256), lds_bytes, s, b, sp);
  return VILO_OK;
}
