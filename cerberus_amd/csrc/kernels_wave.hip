// Single-wave solver for gfx950: what ceres::Solve does per linearisation for Estimator::optimization()
// (estimator.cpp:1221-1236; DENSE_SCHUR + traditional DOGLEG, Ceres 1.14 semantics), one WAVE per window.
//
//   k_assemble_pose   throughput kernel, one thread per entry of the window's 80 x 80 pose / extrinsic / td system: gathers the
//                     per-(start frame, t) Gram slots of k_visual_linearize, the pose blocks of the IMU factor Grams and the prior
//                     image into 15 lower 16 x 16 tiles stored in FP64-MFMA accumulator order (one coalesced load per register in
//                     the solver), plus the gradient of those 80 dimensions.
//   k_solve_wave      one 64-lane workgroup per window, 40 KB of LDS: four windows per CU, one per SIMD, no workgroup barriers and no
//                     idle waves (the four-wave k_build_solve kept one wave of four busy through its serial phases).
//                     Jacobi scaling / dogleg diagonal / q = |J D^-2 g|^2  ->  block-tridiagonal Cholesky chain of the speed / leg-bias
//                     part with its coupling rows T(k) and C -= T^T T on the FP64 matrix cores  ->  landmark Schur complement on the
//                     matrix cores straight from global memory  ->  blocked 80 x 80 Cholesky  ->  triangular solves  ->  bias and landmark
//                     back-substitution  ->  dogleg step and candidate state.
#include "solve_common.hpp"

using namespace vilo;

// packed upper triangle of an IMU factor's 39 x 39 Gram [J | r]^T [J | r], any argument order
__device__ __forceinline__ int ig_idx(int a, int b) { return a <= b ? tri39(a, b) : tri39(b, a); }

// =================================================================================================
// k_assemble_pose
// =================================================================================================
#define CIMG_N 3840   // 15 tiles x 4 registers x 64 lanes
__device__ __forceinline__ int tile_index(int I, int J) { return (I * (I + 1)) / 2 + J; }   // I >= J

// local Gram column (12 .. 24) of a non-pose camera dimension (ex0 66..71, ex1 72..77, td 78)
__device__ __forceinline__ int rest_col(int cd) { return 12 + (cd - CD_EX0); }

// visual part of pose-system entry (hi, lo), hi >= lo, both < 79 (or the gradient: hi = -1): sums the window's Gram slots in chunk
// order, t ascending — a fixed order, so a window gives bitwise the same system whatever batch it is solved in
__device__ double visual_entry(const double *gs, const unsigned *chunk_tab, int nch, int hi, int lo) {
  const bool grad = hi < 0;
  const bool lo_pose = lo < 66;
  const int fl = lo_pose ? lo / 6 : -1, il = lo_pose ? lo - 6 * fl : 0;
  const bool hi_pose = !grad && hi < 66;
  const int fh = hi_pose ? hi / 6 : -1, ih = hi_pose ? hi - 6 * fh : 0;
  const int ch_ = grad ? 25 : (hi_pose ? 0 : rest_col(hi));   // local column of a non-pose hi (or r)
  double sum = 0.0;
  for (int c = 0; c < nch; ++c) {
    const unsigned ct = chunk_tab[c];
    const int s = ct & 255, km = (ct >> 8) & 255, sl0 = ct >> 16;
    if (!lo_pose) {   // rest x rest (or rest x r): every slot of the window
      const int e = tri26(rest_col(lo), ch_);
      for (int t = 0; t < km; ++t) sum += gs[(size_t)(sl0 + t) * VILO_GRAM + e];
    } else if (!hi_pose) {   // pose f x rest / r: as start pose in every slot of its own chunks, as pose j in slot f - s of earlier chunks
      if (s == fl) {
        const int e = tri26(il, ch_);
        for (int t = 0; t < km; ++t) sum += gs[(size_t)(sl0 + t) * VILO_GRAM + e];
      } else if (s < fl && fl - s < km) {
        sum += gs[(size_t)(sl0 + fl - s) * VILO_GRAM + tri26(6 + il, ch_)];
      }
    } else if (fl == fh) {   // pose f x pose f
      if (s == fl) {
        const int e = tri26(il, ih);
        for (int t = 0; t < km; ++t) sum += gs[(size_t)(sl0 + t) * VILO_GRAM + e];
      } else if (s < fl && fl - s < km) {
        sum += gs[(size_t)(sl0 + fl - s) * VILO_GRAM + tri26(6 + il, 6 + ih)];
      }
    } else {   // pose fl x pose fh, fl < fh: start pose fl, observed in frame fh
      if (s == fl && fh - fl < km) sum += gs[(size_t)(sl0 + fh - fl) * VILO_GRAM + tri26(il, 6 + ih)];
    }
  }
  return sum;
}

// IMU-factor part of pose-system entry (hi, lo) (pose x pose only; gradient: hi = -1)
__device__ double imu_pose_entry(const double *igram, int F, int hi, int lo) {
  if (lo >= 66) return 0.0;
  const int fl = lo / 6, il = lo - 6 * fl;
  double sum = 0.0;
  if (hi < 0) {
    if (fl < F - 1) sum += igram[fl * 780 + tri39(il, 38)];
    if (fl >= 1 && fl < F) sum += igram[(fl - 1) * 780 + tri39(19 + il, 38)];
    return sum;
  }
  if (hi >= 66) return 0.0;
  const int fh = hi / 6, ih = hi - 6 * fh;
  if (fh == fl) {
    if (fl < F - 1) sum += igram[fl * 780 + tri39(il, ih)];
    if (fl >= 1 && fl < F) sum += igram[(fl - 1) * 780 + tri39(19 + il, 19 + ih)];
  } else if (fh == fl + 1 && fh < F) {
    sum += igram[fl * 780 + tri39(il, 19 + ih)];
  }
  return sum;
}

#define ASM_THREADS 256
#define ASM_BLOCKS_PER_WIN 16   // 16 x 256 = 4096 >= 3840 matrix entries + 224 gradient entries

__global__ void __launch_bounds__(ASM_THREADS) k_assemble_pose(BatchDev b) {
  __shared__ unsigned chunk_tab[64];
  __shared__ short inv_pmap[CD_N];
  const int win = blockIdx.x / ASM_BLOCKS_PER_WIN, part = blockIdx.x % ASM_BLOCKS_PER_WIN;
  const SolverState &st = b.st[win];
  if (st.done || !st.need_lin) return;
  const WinMeta wm = b.win[win];
  const int tid = threadIdx.x, F = wm.n_frames, nch = min(wm.n_chunks, 64);
  if (tid < nch) {
    const ChunkMeta cm = b.chunk[wm.chunk_off + tid];
    chunk_tab[tid] = (unsigned)cm.s | ((unsigned)cm.kmax << 8) | ((unsigned)(cm.gram_off - wm.gram_off) << 16);
  }
  const double *gs = b.gram + (size_t)wm.gram_off * VILO_GRAM;
  const double *igram = b.imu_gram + (size_t)win * 10 * 780;
  const double *pd = b.prior_dense + (size_t)win * PD_N;
  const int idx = part * ASM_THREADS + tid;
  if (idx >= CIMG_N) {
    // gradient of the 224 camera dimensions: the prior's b0 + H dx for all of them (H dx of the current point was formed by k_accept),
    // the visual and IMU parts for the pose system (the speed / leg-bias part adds its IMU terms in the solver)
    for (int e = tid; e < CD_N; e += ASM_THREADS) inv_pmap[e] = -1;
    __syncthreads();
    if (tid < wm.prior_n) inv_pmap[b.prior_map[(size_t)win * 96 + tid]] = (short)tid;
    __syncthreads();
    const int cd = idx - CIMG_N;
    if (cd < CD_N) {
      const int pi = inv_pmap[cd];
      double g = (wm.prior_n > 0 && pi >= 0) ? b.prior_b0[(size_t)win * 96 + pi] + b.prior_hd[(size_t)win * 96 + pi] : 0.0;
      if (cd < VILO_NPU) g += visual_entry(gs, chunk_tab, nch, -1, cd) + imu_pose_entry(igram, F, -1, cd);
      if (!cd_active(cd, F, wm.const_mask)) g = 0.0;
      b.cam_gin[(size_t)win * CD_N + cd] = g;
    }
    return;
  }
  __syncthreads();
  const int tile = idx >> 8, r = (idx >> 6) & 3, lane = idx & 63, lr = lane & 15, lk = lane >> 4;
  int I = 0;
  while (tile_index(I + 1, 0) <= tile) ++I;
  const int J = tile - tile_index(I, 0);
  const int row = 16 * I + lk + 4 * r, col = 16 * J + lr;
  const int hi = max(row, col), lo = min(row, col);
  double v;
  if (!cd_active(hi, F, wm.const_mask) || !cd_active(lo, F, wm.const_mask)) {
    v = (hi == lo) ? 1.0 : 0.0;   // constant blocks / absent frames / padding: identity rows and columns
  } else {
    v = pd[PD_C + hi * PD_CLD + lo] + visual_entry(gs, chunk_tab, nch, hi, lo) + imu_pose_entry(igram, F, hi, lo);
  }
  b.Cimg[(size_t)win * CIMG_N + idx] = v;
}

// =================================================================================================
// k_solve_wave
// =================================================================================================
// LDS map (doubles; 5120 = 40960 B per workgroup, four workgroups per CU)
#define WS_C 0          // 15 lower tiles, 256 each, element (r, c) of a tile at 16 r + ((c + r) & 15): conflict-free for accumulator-order
                        // accesses (row lk + 4 reg, column lr) and for operand-order accesses (row lr, column 4 kk + lk)
#define WS_G 3840
#define WS_DH2 3920
#define WS_Y 4000
#define WS_V 4080
#define WS_SCR 4160
#define WS_TOTAL 5120
// while the pose tiles live in registers (until the Cholesky) their LDS region holds the speed / leg-bias part
#define WC_AD 0         // [11][169] diagonal blocks
#define WC_AO 1859      // [10][169] rows frame k + 1, columns frame k
#define WC_VB 3552      // [143] v = g / dhat^2 of the speed / leg-bias dimensions
#define WC_DB 3696      // [143] dhat^2
// scratch during the chain
#define WX_LM 0         // 13 x 13: L_k, then M_k = L_k^-1
#define WX_TA0 176
#define WX_TA1 352
#define WX_RINV 528     // 16
#define WX_SN 544       // 13 x 13: S_{k-1} = A_{k-1,k-1} - T_A(k)^T T_A(k)
#define WX_GB 720       // [143] gradient of the speed / leg-bias dimensions
// scratch during the Cholesky / solves
#define WX_D16 0        // 16 x 17
#define WX_LI16 272     // 16 x 17 + 16
#define WX_P16 560      // 16 x 17
#define WX_COL 832      // [80] reciprocal diagonal of the factor
// back-substitution of the speed / leg-bias part (the factor in the C region is dead by then)
#define WB_M 0          // [11][169]
#define WB_TA 1859      // [11][169]
#define WX_U 0          // [143]
#define WX_YB 144       // [143]
#define WX_DEL 288      // [224] step of the camera dimensions
#define WX_GB2 512      // [143] gradient of the speed / leg-bias dimensions (again: the Cholesky scratch overwrote WX_GB)

extern "C" size_t vilo_solve_wave_lds_bytes() { return (size_t)WS_TOTAL * sizeof(double); }

__device__ __forceinline__ int cswz(int t, int r, int c) { return WS_C + 256 * t + 16 * r + ((c + r) & 15); }

// IMU + prior coupling of speed / leg-bias dimension i of frame k with pose-system column p (the B block of the arrow system).
// Factor k (frames k, k + 1) has frame k as "i" (bias rows 6 + i) and factor k - 1 has it as "j" (bias rows 25 + i).
__device__ __forceinline__ double b_coupling(const double *igram, const double *pd, int F, int kb, int cmask, int k, int i, int p) {
  if (i >= 13 || p >= VILO_NPU || !cd_active(CD_B0 + 13 * k + i, F, cmask) || !cd_active(p, F, cmask)) return 0.0;
  double v = 0.0;
  if (p < 66) {
    const int f = p / 6, c = p - 6 * f;
    if (f == k) {
      if (k < F - 1) v += igram[k * 780 + tri39(c, 6 + i)];
      if (k >= 1) v += igram[(k - 1) * 780 + tri39(19 + c, 25 + i)];
    } else if (f == k + 1) {
      if (k < F - 1) v += igram[k * 780 + tri39(6 + i, 19 + c)];
    } else if (f == k - 1) {
      v += igram[(k - 1) * 780 + tri39(c, 25 + i)];
    }
  }
  if (k == kb) v += pd[PD_BP + i * 80 + p];
  return v;
}

// 16 x 16 Cholesky + inverse of the factor by one wave (diagonal tile of the blocked 80 x 80 factorisation).
// A: LDS 16 x 17 row-major in. Lane i (< 16, replicated in the four 16-lane groups) owns row i; pivots broadcast with v_readlane.
// Writes L (lower, zeros above) into swizzled tile t of the C region and L^-1 (lower) to Linv (16 x 17). Returns 0 / 1 (not positive definite).
__device__ __forceinline__ int chol16_tile(double *lds, const double *A, int t, double *Linv) {
  const int lane = threadIdx.x & 63;
  const int row = lane & 15;
  double a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = A[row * 17 + j];
  int fail = 0;
  double myrinv = 1.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double piv = readlane_d(a[j], j);
    if (!(piv > 0.0) || !isfinite(piv)) { fail = 1; piv = 1.0; }
    const double rinv = rsqrt(piv);
    const double lj = (row == j) ? piv * rinv : (row > j ? a[j] * rinv : 0.0);
    a[j] = lj;
    if (row == j) myrinv = rinv;
#pragma unroll
    for (int q = j + 1; q < 16; ++q) a[q] -= lj * readlane_d(lj, q);
  }
  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) lds[cswz(t, lane, j)] = a[j];
    Linv[16 * 17 + lane] = myrinv;
  }
  // column c = lane of L^-1 by forward substitution; L is broadcast from the owning lanes' registers
  double rv[16], cl[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) rv[i] = readlane_d(myrinv, i);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    double v = (i == row) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < i; ++q) v -= readlane_d(a[q], i) * cl[q];   // L[i][q] lives in lane i, register q
    cl[i] = v * rv[i];
  }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) Linv[i * 17 + lane] = (i >= lane) ? cl[i] : 0.0;
  }
  lds_fence();
  return fail;
}

__global__ void __launch_bounds__(64) k_solve_wave(BatchDev b, SolveParams sp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int win = blockIdx.x;
  SolverState &st = b.st[win];
  if (st.done) return;
  const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
  const WinMeta wm = b.win[win];
  const int F = wm.n_frames, L = wm.L, kb = wm.pad, cmask = wm.const_mask;
  double *x = b.x + (size_t)win * XSTRIDE, *xc = b.xc + (size_t)win * XSTRIDE;
  double *cam_g = b.cam_g + (size_t)win * CD_N, *cam_dh2 = b.cam_dh2 + (size_t)win * CD_N, *cam_y = b.cam_y + (size_t)win * CD_N;
  double *cam_scale = b.cam_scale + (size_t)win * CD_N;
  double *g = lds + WS_G, *dh2 = lds + WS_DH2, *y = lds + WS_Y, *v = lds + WS_V, *scr = lds + WS_SCR;
  // gradient, dogleg diagonal and Gauss-Newton step of the 143 speed / leg-bias dimensions: dimension lane + 64 m in register m
  double gBr[3], dBr[3], yBr[3];

  if (st.need_lin) {
    if (lane == 0) st.phase_clk[0] = clock64();
    const double *Cimg = b.Cimg + (size_t)win * CIMG_N;
    const double *gin = b.cam_gin + (size_t)win * CD_N;
    const double *igram = b.imu_gram + (size_t)win * 10 * 780;
    const double *pd = b.prior_dense + (size_t)win * PD_N;
    const double *wl = b.lm_w + 80 * (size_t)wm.lm_off;
    double *lm_E = b.lm_E + wm.lm_off, *lm_g = b.lm_g + wm.lm_off, *lm_dh2 = b.lm_dh2 + wm.lm_off, *lm_scale = b.lm_scale + wm.lm_off,
           *lm_einv = b.lm_einv + wm.lm_off, *lm_y = b.lm_y + wm.lm_off;
    double *Mg = b.Lk + (size_t)win * 11 * 169, *TAg = b.TAg + (size_t)win * 11 * 169;
    const bool first_scale = !st.scale_ready;

    // ---- the speed / leg-bias part into LDS: diagonal blocks A_kk (IMU factors k and k - 1, prior at frame kb), off-diagonal blocks
    //      A_{k+1,k}; constant / absent dimensions as identity rows ----
    for (int e = lane; e < 11 * 169; e += 64) {
      const int k = e / 169, ij = e - 169 * k, i = ij / 13, j = ij - 13 * i;
      double val;
      if (!cd_active(CD_B0 + 13 * k + i, F, cmask) || !cd_active(CD_B0 + 13 * k + j, F, cmask)) {
        val = (i == j) ? 1.0 : 0.0;
      } else {
        val = (k == kb) ? pd[PD_AD + e] : 0.0;
        if (k < F - 1) val += igram[k * 780 + ig_idx(6 + i, 6 + j)];
        if (k >= 1) val += igram[(k - 1) * 780 + ig_idx(25 + i, 25 + j)];
      }
      lds[WC_AD + e] = val;
    }
    for (int e = lane; e < 10 * 169; e += 64) {
      const int k = e / 169, ij = e - 169 * k, i = ij / 13, j = ij - 13 * i;   // row: dimension i of frame k + 1, column: dimension j of frame k
      double val = 0.0;
      if (k < F - 1 && cd_active(CD_B0 + 13 * (k + 1) + i, F, cmask) && cd_active(CD_B0 + 13 * k + j, F, cmask)) val = igram[k * 780 + tri39(6 + j, 25 + i)];
      lds[WC_AO + e] = val;
    }
    // gradient, Jacobi scaling (first linearisation), dogleg diagonal and v = D^-2 g of the speed / leg-bias dimensions
    double part_gn = 0.0, part_gmax = 0.0, part_q = 0.0;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int e = lane + 64 * m;
      gBr[m] = 0.0; dBr[m] = 1.0; yBr[m] = 0.0;
      if (e < 143) {
        const int k = e / 13, i = e - 13 * k, cd = CD_B0 + e;
        double hii = 1.0, ge = 0.0, d = 1.0, ve = 0.0;
        if (cd_active(cd, F, cmask)) {
          hii = (k == kb) ? pd[PD_AD + k * 169 + i * 14] : 0.0;
          ge = gin[cd];
          if (k < F - 1) { hii += igram[k * 780 + tri39(6 + i, 6 + i)]; ge += igram[k * 780 + tri39(6 + i, 38)]; }
          if (k >= 1) { hii += igram[(k - 1) * 780 + tri39(25 + i, 25 + i)]; ge += igram[(k - 1) * 780 + tri39(25 + i, 38)]; }
          double sc;
          if (first_scale) { sc = sp.jacobi_scaling ? 1.0 / (1.0 + sqrt(hii)) : 1.0; cam_scale[cd] = sc; }
          else sc = cam_scale[cd];
          const double d2 = fmin(fmax(sc * sc * hii, sp.min_lm_diagonal), sp.max_lm_diagonal);
          d = d2 / (sc * sc);
          ve = ge / d;
        }
        gBr[m] = ge; dBr[m] = d;
        lds[WC_VB + e] = ve;
        lds[WC_DB + e] = d;
        scr[WX_GB + e] = ge;
        part_gn += ge * ve;
        part_gmax = fmax(part_gmax, fabs(ge));
      }
    }

    bool solved = false;
    bool have_q = false;
    double gnorm2 = 0.0, gmax = 0.0, qq = 0.0, gnnorm2 = 0.0, gy = 0.0;
    double mu = st.mu;
    while (!solved) {
      if (lane == 0) st.phase_clk[1] = clock64();
      // ---- pose system: 15 lower tiles in accumulator order, one coalesced load per register ----
      mfma_d4 acc[15];
#pragma unroll
      for (int t = 0; t < 15; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = Cimg[(t * 4 + r) * 64 + lane];
      if (!have_q) {
        // diagonal -> Jacobi scaling, dogleg diagonal, v = D^-2 g
#pragma unroll
        for (int I = 0; I < 5; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (lk + 4 * r == lr) y[16 * I + lr] = acc[tile_index(I, I)][r];
        lds_fence();
        for (int cd = lane; cd < 80; cd += 64) {
          const double ge = gin[cd];
          double d = 1.0, ve = 0.0;
          if (cd_active(cd, F, cmask)) {
            const double hii = y[cd];
            double sc;
            if (first_scale) { sc = sp.jacobi_scaling ? 1.0 / (1.0 + sqrt(hii)) : 1.0; cam_scale[cd] = sc; }
            else sc = cam_scale[cd];
            const double d2 = fmin(fmax(sc * sc * hii, sp.min_lm_diagonal), sp.max_lm_diagonal);
            d = d2 / (sc * sc);
            ve = ge / d;
          }
          g[cd] = ge; dh2[cd] = d; v[cd] = ve;
          part_gn += ge * ve;
          part_gmax = fmax(part_gmax, fabs(ge));
        }
        lds_fence();
        // q = v^T H v: pose tiles
#pragma unroll
        for (int I = 0; I < 5; ++I)
#pragma unroll
          for (int J = 0; J <= I; ++J) {
            const double vc = v[16 * J + lr], sym = (I == J) ? 1.0 : 2.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) part_q += sym * v[16 * I + lk + 4 * r] * acc[tile_index(I, J)][r] * vc;
          }
        // speed / leg-bias rows: v_B^T (A_BB v_B + 2 B v_P), one lane per (frame, dimension)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const int e = lane + 64 * m;
          if (e < 143) {
            const int k = e / 13, i = e - 13 * k;
            if (k < F) {
              const double vi = lds[WC_VB + e];
              double sacc = 0.0, cross = 0.0, bp = 0.0;
              for (int j = 0; j < 13; ++j) sacc += lds[WC_AD + k * 169 + i * 13 + j] * lds[WC_VB + 13 * k + j];
              if (k > 0)
                for (int j = 0; j < 13; ++j) cross += lds[WC_AO + (k - 1) * 169 + i * 13 + j] * lds[WC_VB + 13 * (k - 1) + j];
              if (vi != 0.0) {   // (an inactive dimension has v = 0 and no coupling)
                const int p0 = (k == kb) ? 0 : max(0, 6 * (k - 1)), p1 = (k == kb) ? VILO_NPU : min(66, 6 * (k + 2));
                for (int p = p0; p < p1; ++p) bp += b_coupling(igram, pd, F, kb, cmask, k, i, p) * v[p];
              }
              part_q += vi * (sacc + 2.0 * cross + 2.0 * bp);
            }
          }
        }
        // landmarks, pass 1
        for (int l = lane; l < L; l += 64) {
          const double E = lm_E[l], gl = lm_g[l];
          double sc;
          if (first_scale) { sc = sp.jacobi_scaling ? 1.0 / (1.0 + sqrt(E)) : 1.0; lm_scale[l] = sc; }
          else sc = lm_scale[l];
          const double d2 = fmin(fmax(sc * sc * E, sp.min_lm_diagonal), sp.max_lm_diagonal) / (sc * sc);
          lm_dh2[l] = d2;
          const double vl = gl / d2;
          part_q += E * vl * vl;   // the cross term 2 vl w_l^T v is accumulated in the Schur pass below
          lm_y[l] = vl;            // (scratch until the back-substitution overwrites it)
          part_gn += gl * vl;
          part_gmax = fmax(part_gmax, fabs(gl));
        }
        gnorm2 = wave_sum(part_gn);
        gmax = wave_max(part_gmax);
        if (!sp.fixed_iterations && gmax <= sp.gradient_tolerance) {
          if (lane == 0) { st.gmax = gmax; st.done = 1; st.termination = 1; st.step_valid = 0; }
          return;
        }
      }
      for (int l = lane; l < L; l += 64) lm_einv[l] = 1.0 / (lm_E[l] + mu * lm_dh2[l]);
      if (lane == 0) st.phase_clk[2] = clock64();

      // ---- block-tridiagonal Cholesky chain of the speed / leg-bias part (13 x 13 blocks, frames F-1 .. 0):
      //        S_k = A_kk + mu D_k - T_A(k+1)^T T_A(k+1),  L_k = chol(S_k),  M_k = L_k^-1,  T_A(k) = M_k A_{k,k-1},
      //        V = [B_k | g_k] - T_A(k+1)^T T(k+1),  T(k) = M_k V   (13 x 80: columns 0..78 coupling rows, column 79 = rhs),
      //        C -= T_B(k)^T T_B(k),  rhs_P -= T_B(k)^T t_g(k).
      //      The scalar part runs lane = row in the four 16-lane groups; T, V and the rank update are FP64-MFMA tiles whose accumulator
      //      layout (register r of lane (lr, lk) = row lk + 4 r, column lr) is the operand layout of the next product. ----
      int fail = 0;
      {
        const int grp = lk, c = lr;
        const int row = c < 13 ? c : 0;
        double *LM = scr + WX_LM, *rinvk = scr + WX_RINV, *SN = scr + WX_SN, *GB = scr + WX_GB;
        double *TAcur = scr + WX_TA0, *TAprev = scr + WX_TA1;
        mfma_d4 T[5];
        double yr[5];
#pragma unroll
        for (int X = 0; X < 5; ++X) { T[X] = mfma_d4{0.0, 0.0, 0.0, 0.0}; yr[X] = 0.0; }
        for (int k = F - 1; k >= 0; --k) {
          // [B_k | g_k] in accumulator order: row lk + 4 r (< 13), column 16 X + lr; column 79 carries the gradient
          mfma_d4 V[5];
          const int x_lo = max(0, (6 * (k - 1)) >> 4), x_hi = min(4, (6 * (k + 2) - 1) >> 4);
#pragma unroll
          for (int X = 0; X < 5; ++X) {
            V[X] = mfma_d4{0.0, 0.0, 0.0, 0.0};
            if (k == kb || (X >= x_lo && X <= x_hi)) {
#pragma unroll
              for (int r = 0; r < 4; ++r) V[X][r] = b_coupling(igram, pd, F, kb, cmask, k, lk + 4 * r, 16 * X + lr);
            }
          }
          if (lr == 15) {
#pragma unroll
            for (int r = 0; r < 4; ++r) V[4][r] = (lk + 4 * r < 13) ? GB[13 * k + lk + 4 * r] : 0.0;
          }
          // S_k (lane = row): first frame straight from A_kk, later frames from the update left by the previous step
          double a[13], l[13];
          const double *Ssrc = (k == F - 1) ? lds + WC_AD + k * 169 : SN;
#pragma unroll
          for (int j = 0; j < 13; ++j) { a[j] = Ssrc[row * 13 + j]; l[j] = 0.0; }
          {
            const double md = mu * lds[WC_DB + 13 * k + row];
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
          double rhs[13], cl[13], rv[13];
#pragma unroll
          for (int i = 0; i < 13; ++i) {
            rv[i] = readlane_d(myrinv, i);
            if (grp == 0) rhs[i] = (k > 0) ? lds[WC_AO + max(k - 1, 0) * 169 + i * 13 + row] : 0.0;
            else rhs[i] = (i == c) ? 1.0 : 0.0;
          }
#pragma unroll
          for (int i = 0; i < 13; ++i) {
            double vv = rhs[i];
#pragma unroll
            for (int q = 0; q < i; ++q) vv -= readlane_d(l[q], i) * cl[q];
            cl[i] = vv * rv[i];
          }
          if (c < 13 && grp < 2) {
            if (grp == 0) {
#pragma unroll
              for (int i = 0; i < 13; ++i) { TAcur[i * 13 + c] = cl[i]; TAg[k * 169 + i * 13 + c] = cl[i]; }
            } else {
#pragma unroll
              for (int i = 0; i < 13; ++i) { LM[i * 13 + c] = cl[i]; Mg[k * 169 + i * 13 + c] = cl[i]; }
            }
          }
          lds_fence();
          // S_{k-1} = A_{k-1,k-1} - T_A(k)^T T_A(k)
          if (k > 0) {
            for (int e = lane; e < 169; e += 64) {
              const int i = e / 13, j = e - 13 * i;
              double sacc = 0.0;
#pragma unroll
              for (int q = 0; q < 13; ++q) sacc += TAcur[q * 13 + i] * TAcur[q * 13 + j];
              SN[e] = lds[WC_AD + (k - 1) * 169 + e] - sacc;
            }
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
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) V[X] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[kk], T[X][kk], V[X], 0, 0, 0);
          }
#pragma unroll
          for (int X = 0; X < 5; ++X) {
            mfma_d4 n = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) n = __builtin_amdgcn_mfma_f64_16x16x4f64(am[kk], V[X][kk], n, 0, 0, 0);
            T[X] = n;
          }
          // t_g(k) (column 79) to every lane of its 16-lane row group; the pose system must not see it
          mfma_d4 T4 = T[4];
          double tg[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            tg[kk] = __shfl(T[4][kk], (lane & 48) | 15, 64);
            if (lr == 15) T4[kk] = 0.0;
          }
          // C -= T_B^T T_B, rhs_P -= T_B^T t_g
#pragma unroll
          for (int I = 0; I < 5; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J)
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) {
                const double opa = (I == 4) ? T4[kk] : T[I][kk], opb = (J == 4) ? T4[kk] : T[J][kk];
                acc[tile_index(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-opa, opb, acc[tile_index(I, J)], 0, 0, 0);
              }
#pragma unroll
          for (int X = 0; X < 5; ++X)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) yr[X] += ((X == 4) ? T4[kk] : T[X][kk]) * tg[kk];
          double *sw = TAcur; TAcur = TAprev; TAprev = sw;
          lds_fence();
        }
        // reduced right-hand side so far: g_P - sum_k T_B^T t_g
#pragma unroll
        for (int X = 0; X < 5; ++X) {
          yr[X] += __shfl_xor(yr[X], 16, 64);
          yr[X] += __shfl_xor(yr[X], 32, 64);
          if (lk == 0) v[16 * X + lr] = g[16 * X + lr] - yr[X];
        }
      }
      if (lane == 0) st.phase_clk[3] = clock64();

      // ---- Schur complement of the landmarks on the FP64 matrix cores: C -= sum_l w_l w_l^T / (E_l + mu dhat_l^2). One k-step = 4
      //      landmarks; the operand of tile row X (lane: w[16 X + lr][4 kk + lk]) serves as A of tiles (X, .) and as B of tiles (., X):
      //      5 row-coalesced global loads and 15 MFMAs per k-step, no LDS. The same operands give rhs_P -= sum_l w_l g_l / (...) and the
      //      2 v_l w_l^T v_P term of q. ----
      {
        double actv[5], vv[5], yacc[5], qacc = 0.0;
#pragma unroll
        for (int X = 0; X < 5; ++X) {
          actv[X] = cd_active(16 * X + lr, F, cmask) ? 1.0 : 0.0;
          // v_P for the cross term of q (the LDS vector v holds the reduced right-hand side by now)
          vv[X] = (actv[X] != 0.0) ? g[16 * X + lr] / dh2[16 * X + lr] : 0.0;
          yacc[X] = 0.0;
        }
        const int nks = (L + 3) >> 2;
        double opb[2][4][5], eb[2][4], gb[2][4], db[2][4];
        auto ldtrip = [&](int kk0, int bsel) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int l = 4 * (kk0 + u) + lk, lc = min(l, L - 1);
            eb[bsel][u] = lm_einv[lc]; gb[bsel][u] = lm_g[lc]; db[bsel][u] = lm_y[lc];
#pragma unroll
            for (int X = 0; X < 5; ++X) opb[bsel][u][X] = wl[(size_t)(16 * X + lr) * L + lc];
          }
        };
        auto dotrip = [&](int kk0, int bsel) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int l = 4 * (kk0 + u) + lk;
            const double ei = (l < L) ? eb[bsel][u] : 0.0, ge = gb[bsel][u] * ei, vl = (l < L) ? db[bsel][u] : 0.0;
            double op[5];
#pragma unroll
            for (int X = 0; X < 5; ++X) op[X] = opb[bsel][u][X] * actv[X];
#pragma unroll
            for (int I = 0; I < 5; ++I)
#pragma unroll
              for (int J = 0; J <= I; ++J)
                acc[tile_index(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-(op[I] * ei), op[J], acc[tile_index(I, J)], 0, 0, 0);
#pragma unroll
            for (int X = 0; X < 5; ++X) { yacc[X] += op[X] * ge; qacc += op[X] * vv[X] * vl; }
          }
        };
        if (L > 0) {
          // the landmark vectors written above (lm_einv, lm_y) are read back through global memory by other lanes
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          ldtrip(0, 0);
          for (int kk0 = 0; kk0 < nks; kk0 += 8) {
            ldtrip(kk0 + 4, 1);
            dotrip(kk0, 0);
            ldtrip(kk0 + 8, 0);
            dotrip(kk0 + 4, 1);
          }
        }
        if (!have_q) {
          part_q += 2.0 * qacc;
          qq = wave_sum(part_q);
          have_q = true;
        }
#pragma unroll
        for (int X = 0; X < 5; ++X) {
          yacc[X] += __shfl_xor(yacc[X], 16, 64);
          yacc[X] += __shfl_xor(yacc[X], 32, 64);
          if (lk == 0) v[16 * X + lr] -= yacc[X];
        }
        // regularise: diag += mu dhat^2
#pragma unroll
        for (int I = 0; I < 5; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (lk + 4 * r == lr) acc[tile_index(I, I)][r] += mu * dh2[16 * I + lr];
      }
      lds_fence();
      if (lane == 0) st.phase_clk[4] = clock64();

      // ---- dense Cholesky of the 80 x 80 reduced pose system, blocked by 16: diagonal tile in registers + v_readlane (also its
      //      inverse), panel L_Ij = A_Ij L_jj^-T and trailing update A_IJ -= L_Ij L_Jj^T on the FP64 matrix cores. L is left in the C
      //      region (swizzled lower tiles) for the solves ----
      {
        double *D16 = scr + WX_D16, *LI16 = scr + WX_LI16, *P16 = scr + WX_P16;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
          for (int r = 0; r < 4; ++r) D16[(lk + 4 * r) * 17 + lr] = acc[tile_index(j, j)][r];
          lds_fence();
          fail |= chol16_tile(lds, D16, tile_index(j, j), LI16);
          double li[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) li[kk] = LI16[lr * 17 + 4 * kk + lk];
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
            for (int r = 0; r < 4; ++r) lds[cswz(t, lk + 4 * r, lr)] = nacc[r];
            lds_fence();   // (P16 is reused by the next panel)
          }
          // trailing update: operand (row lr, columns 4 kk + lk) of every panel tile once
          double pa[5][4];
#pragma unroll
          for (int I = j + 1; I < 5; ++I)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) pa[I][kk] = lds[cswz(tile_index(I, j), lr, 4 * kk + lk)];
#pragma unroll
          for (int I = j + 1; I < 5; ++I)
#pragma unroll
            for (int J = j + 1; J <= I; ++J)
#pragma unroll
              for (int kk = 0; kk < 4; ++kk)
                acc[tile_index(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[I][kk], pa[J][kk], acc[tile_index(I, J)], 0, 0, 0);
        }
      }
      if (fail) {
        // DoglegStrategy::ComputeGaussNewtonStep: mu *= 10 and retry while mu < max_mu (1.0)
        mu *= 10.0;
        if (lane == 0) st.mu = mu;
        if (!(mu < 1.0)) {
          if (lane == 0) { st.lin_fail = 1; st.step_valid = 0; st.gnorm2 = gnorm2; st.q = qq; st.gmax = gmax; st.scale_ready = 1; }
          return;
        }
        continue;
      }
      if (lane == 0) st.phase_clk[5] = clock64();

      // ---- L L^T yP = rhs: lane owns rows lane and lane + 64; pivots by v_readlane; the factor is read in blocks of 16 columns into
      //      registers so that the 160 dependent steps touch no memory ----
      {
        double *col = scr + WX_COL;
        for (int cd = lane; cd < 80; cd += 64) col[cd] = 1.0 / lds[cswz(tile_index(cd >> 4, cd >> 4), cd & 15, cd & 15)];
        lds_fence();
        double b0 = v[lane], b1 = lane < 16 ? v[lane + 64] : 0.0;
        const int I0 = lane >> 4;   // tile row of this lane's first row; its second row (lane + 64 < 80) is in tile row 4
#pragma unroll
        for (int jb = 0; jb < 5; ++jb) {
          double l0[16], l1[16], ri[16];
#pragma unroll
          for (int jj = 0; jj < 16; ++jj) {
            l0[jj] = (I0 >= jb) ? lds[cswz(tile_index(max(I0, jb), jb), lane & 15, jj)] : 0.0;
            l1[jj] = lds[cswz(tile_index(4, jb), lane & 15, jj)];
            ri[jj] = col[16 * jb + jj];
          }
#pragma unroll
          for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * jb + jj;
            const double yj = readlane_d((jb < 4) ? b0 : b1, j & 63) * ri[jj];
            if (jb < 4 && lane > j) b0 -= l0[jj] * yj;
            if (lane < 16 && lane + 64 > j) b1 -= l1[jj] * yj;
            if (lane == (j & 63)) { if (jb < 4) b0 = yj; else b1 = yj; }
          }
        }
#pragma unroll
        for (int jb = 4; jb >= 0; --jb) {
          double c0[16], c1[16], ri[16];
#pragma unroll
          for (int jj = 0; jj < 16; ++jj) {
            c0[jj] = (jb >= I0) ? lds[cswz(tile_index(jb, min(I0, jb)), jj, lane & 15)] : 0.0;   // L[16 jb + jj][lane]
            c1[jj] = (jb == 4) ? lds[cswz(tile_index(4, 4), jj, lane & 15)] : 0.0;               // L[16 jb + jj][lane + 64]
            ri[jj] = col[16 * jb + jj];
          }
#pragma unroll
          for (int jj = 15; jj >= 0; --jj) {
            const int j = 16 * jb + jj;
            const double yj = readlane_d((jb < 4) ? b0 : b1, j & 63) * ri[jj];
            if (lane < j) b0 -= c0[jj] * yj;
            if (lane < 16 && lane + 64 < j) b1 -= c1[jj] * yj;
            if (lane == (j & 63)) { if (jb < 4) b0 = yj; else b1 = yj; }
          }
        }
        y[lane] = cd_active(lane, F, cmask) ? b0 : 0.0;
        if (lane < 16) y[lane + 64] = cd_active(lane + 64, F, cmask) ? b1 : 0.0;
      }
      lds_fence();
      if (lane == 0) st.phase_clk[6] = clock64();

      // ---- back-substitution of the speed / leg-bias part: c_k = g_k - B_k yP, then the two block-bidiagonal sweeps
      //        u_k = M_k (c_k - T_A(k+1)^T u_{k+1})   k = F-1 .. 0,      y_k = M_k^T (u_k - T_A(k) y_{k-1})   k = 0 .. F-1 ----
      double part_gnn = 0.0, part_gy = 0.0;
      {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // M_k / T_A(k) written by this wave during the chain
        for (int e = lane; e < F * 169; e += 64) { lds[WB_M + e] = Mg[e]; lds[WB_TA + e] = TAg[e]; }
        double *U = scr + WX_U, *YB = scr + WX_YB, *GB2 = scr + WX_GB2;
        // c_k: lane (lr = dimension i, lk = quarter of the columns); the IMU part of B_k spans poses k-1 .. k+1, the prior part frame kb only
        for (int k = 0; k < F; ++k) {
          double sacc = 0.0;
          if (lr < 13) {
            const int p0 = (k == kb) ? 0 : max(0, 6 * (k - 1)), p1 = (k == kb) ? VILO_NPU : min(66, 6 * (k + 2));
            for (int p = p0 + lk; p < p1; p += 4) sacc += b_coupling(igram, pd, F, kb, cmask, k, lr, p) * y[p];
          }
          sacc += __shfl_xor(sacc, 16, 64);
          sacc += __shfl_xor(sacc, 32, 64);
          if (lane < 13) U[13 * k + lane] = sacc;   // (B_k yP)_i
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
          if (lane + 64 * m < 143) GB2[lane + 64 * m] = gBr[m];
        lds_fence();
        const int row = lr < 13 ? lr : 0;
        // forward sweep
        double unext = 0.0;   // u_{k+1}[row]
        for (int k = F - 1; k >= 0; --k) {
          double s = GB2[13 * k + row] - U[13 * k + row];
          if (k < F - 1) {
#pragma unroll
            for (int q = 0; q < 13; ++q) s -= lds[WB_TA + (k + 1) * 169 + q * 13 + row] * readlane_d(unext, q);
          }
          double u = 0.0;
#pragma unroll
          for (int q = 0; q < 13; ++q) u += lds[WB_M + k * 169 + row * 13 + q] * readlane_d(s, q);
          if (lane < 13) U[13 * k + lane] = u;
          unext = u;
        }
        lds_fence();
        // backward sweep
        double yprev = 0.0;
        for (int k = 0; k < F; ++k) {
          double s = U[13 * k + row];
          if (k > 0) {
#pragma unroll
            for (int q = 0; q < 13; ++q) s -= lds[WB_TA + k * 169 + row * 13 + q] * readlane_d(yprev, q);
          }
          double yk = 0.0;
#pragma unroll
          for (int q = 0; q < 13; ++q) yk += lds[WB_M + k * 169 + q * 13 + row] * readlane_d(s, q);
          if (!cd_active(CD_B0 + 13 * k + row, F, cmask)) yk = 0.0;
          if (lane < 13) YB[13 * k + lane] = yk;
          yprev = yk;
        }
        lds_fence();
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const int e = lane + 64 * m;
          yBr[m] = (e < 13 * F) ? YB[e] : 0.0;
          part_gnn += dBr[m] * yBr[m] * yBr[m] * ((e < 143 && cd_active(CD_B0 + e, F, cmask)) ? 1.0 : 0.0);
          part_gy += gBr[m] * yBr[m];
        }
      }
      if (lane == 0) st.phase_clk[7] = clock64();
      // ---- landmarks: y_l = (g_l - w_l^T yP) / (E_l + mu dhat_l^2), all 80 coupling entries of a landmark in flight at once ----
      for (int l = lane; l < L; l += 64) {
        double wcol[80];
#pragma unroll
        for (int a = 0; a < 80; ++a) wcol[a] = wl[(size_t)a * L + l];
        const double gl = lm_g[l], ei = lm_einv[l], d2 = lm_dh2[l];
        double tl = 0.0;
#pragma unroll
        for (int a = 0; a < VILO_NPU; ++a) tl += wcol[a] * y[a];   // y is zero on inactive dimensions
        const double yl = (gl - tl) * ei;
        lm_y[l] = yl;
        part_gnn += d2 * yl * yl;
        part_gy += gl * yl;
      }
      for (int cd = lane; cd < 80; cd += 64) {
        part_gnn += dh2[cd] * y[cd] * y[cd] * (cd_active(cd, F, cmask) ? 1.0 : 0.0);
        part_gy += g[cd] * y[cd];
      }
      gnnorm2 = wave_sum(part_gnn);
      gy = wave_sum(part_gy);
      if (!(isfinite(gnnorm2) && isfinite(gy))) {   // IsArrayValid(gauss_newton_step_) failed
        mu *= 10.0;
        if (lane == 0) st.mu = mu;
        if (!(mu < 1.0)) {
          if (lane == 0) { st.lin_fail = 1; st.step_valid = 0; st.scale_ready = 1; }
          return;
        }
        continue;
      }
      solved = true;
    }
    // keep the linearisation's vectors for the steps that reuse it after a rejected candidate
    for (int cd = lane; cd < 80; cd += 64) { cam_g[cd] = g[cd]; cam_dh2[cd] = dh2[cd]; cam_y[cd] = y[cd]; }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int e = lane + 64 * m;
      if (e < 144) { cam_g[CD_B0 + e] = gBr[m]; cam_dh2[CD_B0 + e] = dBr[m]; cam_y[CD_B0 + e] = yBr[m]; }
    }
    if (lane == 0) {
      st.gnorm2 = gnorm2; st.gnnorm2 = gnnorm2; st.gdotgn = -gy; st.q = qq; st.gmax = gmax;
      st.alpha = gnorm2 / qq;
      st.scale_ready = 1;
      st.lin_fail = 0;
      st.phase_clk[8] = clock64();
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
  if (!go) return;
  double *del = scr + WX_DEL;
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
  if (lane == 0) st.phase_clk[9] = clock64();
}

// =================================================================================================
// launch
// =================================================================================================
int vilo_launch_wave_solver(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s, int stage) {
  const size_t lds_bytes = (size_t)WS_TOTAL * sizeof(double);
  if (stage == 0) {
    hipLaunchKernelGGL(k_assemble_pose, dim3(b.W * ASM_BLOCKS_PER_WIN), dim3(ASM_THREADS), 0, s, b);
  } else {
    if (!ctx->wave_attr_set) {
      VILO_HIP(hipFuncSetAttribute((const void *)k_solve_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      ctx->wave_attr_set = true;
    }
    hipLaunchKernelGGL(k_solve_wave, dim3(b.W), dim3(64), lds_bytes, s, b, sp);
  }
  return VILO_OK;
}
