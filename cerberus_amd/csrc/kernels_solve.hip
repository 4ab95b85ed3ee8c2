// Linearisation, cost and trust-region bookkeeping kernels of the sliding-window solver for gfx950: what ceres::Solve does for
// Estimator::optimization() (estimator.cpp:1054-1245; DENSE_SCHUR + traditional DOGLEG, Ceres 1.14 semantics) as a batched kernel
// pipeline over independent windows. Jacobians are never materialised in HBM:
//
//   k_visual_linearize  one wave per packed wave of landmarks (lane = landmark): Projection*Factor residual blocks + Huber corrector in
//                       registers, the landmark's 1x1 Hessian / gradient / camera coupling row in registers, the camera-side Gram
//                       blocks of every (start, t) pair on the FP64 matrix cores, and the robust cost of the point it evaluates.
//   k_imu_linearize     one wave per IMULegFactor: raw residual + Jacobian into LDS, whitening by the hoisted sqrt_info and the factor's
//                       39 x 39 Gram (its last diagonal entry is the factor's cost) on the FP64 matrix cores.
//   k_assemble / k_solve_wave (kernels_wave.hip)   normal equations, Schur complement, factorisation, dogleg step, candidate state.
//   k_accept            candidate cost, step quality, accept / reject, radius update.
// The candidate of iteration i is evaluated AND linearised by one pass (every accepted candidate is the next linearisation point; a
// rejected one costs a wasted linearisation, which is rare): observations and preintegration records are read once per iteration.
// k_visual_cost / k_imu_cost are the cost-only forms for the last candidate of a solve.
#include <type_traits>
#include "solve_common.hpp"
#include "visual_lin.hpp"
#include "accept_body.hpp"
#include "lin_common.hpp"

using namespace vilo;

// inverse of the packed upper-triangle index: row a of entry e for an n x n matrix (start(a) = a (2n + 1 - a) / 2)
__device__ __forceinline__ int tri_row(int e, int n) {
  const float tn = (float)(2 * n + 1);
  int a = (int)((tn - sqrtf(tn * tn - 8.0f * (float)e)) * 0.5f);
  if (a < 0) a = 0;
  if (a > n - 1) a = n - 1;
  while (a > 0 && (a * (2 * n + 1 - a)) / 2 > e) --a;
  while (((a + 1) * (2 * n - a)) / 2 <= e) ++a;
  return a;
}

// Block b of a launch runs on XCD b % 8 (observed dispatch order); the packed waves of a window differ in length (kmax 11 / 9 / 7 / 5
// at config 2) and repeat with the window period, so the identity mapping hands every XCD waves of ONE length and the XCD with the
// longest ones finishes 1.4x after the average. Rotating the position inside each group of 8 blocks by the group index gives every
// XCD the same mix. A permutation of [0, n): groups of 8 map onto themselves, the ragged tail is left alone.
__device__ __forceinline__ int xcd_balanced(int b, int n) {
  const int q = b >> 3;
  if (8 * q + 8 > n) return b;
  return 8 * q + ((b + q) & 7);
}


// block-wide sum through LDS (all threads must call); red has blockDim.x entries
__device__ double block_sum(double v, double *red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}
__device__ double block_max(double v, double *red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (t < s) red[t] = fmax(red[t], red[t + s]);
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// =================================================================================================
// k_visual_linearize
// =================================================================================================
#define XROW 24   // LDS stride of a row: 23 columns + 1 pad (rows start 16-byte aligned: ds_write_b128)
#define XLANE 50  // LDS stride per lane: 2 rows + 2 pad (8 consecutive lanes of a 16-byte store hit 8 distinct groups of 4 banks)

// TPAR = false: one wave per packed wave walks all its frames (the throughput form: landmark-side sums stay in registers).
// TPAR = true (small batches, b.lm_part != null): one wave per (packed wave, frame offset) so that a handful of windows still
// fills the chip; every landmark-side term is written per (frame, camera) and k_visual_reduce adds the terms in the order the
// walking form adds them — the two forms give bitwise the same linearisation.
// The factor bodies are visual_lin.hpp's: rotation products hoisted per (start frame, observing frame) pair into an LDS table that 48
// lanes of the wave build for VT_TB frames at a time, Huber weight folded into the projection Jacobian.
#define VT_TB 2       // frames per table build: lane = (frame of the pair, segment, camera, row) = 2 x 4 x 2 x 3
// COMPACT (solve passes of batches whose windows all keep td constant; BatchDev::compact): 16-column rows (visual_lin.hpp: GK_*), one
// Gram tile per camera, slots of VILO_GRAMC doubles for k_assemble<true>; everything on the landmark side is the same.
#define XROWC 16   // compact rows: 16 columns, a lane's two rows back to back
#define XLANEC 34  // + 2 pad: the four rows of a k-step start 0 / 32 / 4 / 36 banks apart (conflict-free operand reads, 16-byte aligned stores)
// LDS of one workgroup of the single-wave visual forms (doubles; every array starts 16-byte aligned). The kernels own the pool and hand it
// to the body, so that a kernel whose workgroups take different roles (k_lin_small_c: visual and IMU workgroups in one launch) pays for the
// larger role's LDS, not for the sum.
template <bool COMPACT> struct VisPool {
  static constexpr int XL = COMPACT ? XLANEC : XLANE;
  static constexpr int O_X = 0, N_X = (64 + 4) * XL + 16;          // 4 zero pad lanes = 8 pad rows
  static constexpr int O_XS = O_X + N_X;                            // the window's state: poses are indexed per lane (lanes of a wave have different start frames)
  static constexpr int O_WT = O_XS + XSTRIDE;
  static constexpr int O_TAB = O_WT + ((VW_N + 1) & ~1);
  static constexpr int N = O_TAB + VT_TB * 4 * VT_N;
};
template <bool TPAR, bool COMPACT>
__device__ __forceinline__ void visual_linearize_body(BatchDev &b, double sq, double huber_a, int mode, double *pool, int bx, int by) {
  constexpr int XR = COMPACT ? XROWC : XROW, XL = COMPACT ? XLANEC : XLANE;
  typedef VisPool<COMPACT> VP;
  static_assert((VP::N_X & 1) == 0 && (VP::O_XS & 1) == 0 && (VP::O_WT & 1) == 0 && (VP::O_TAB & 1) == 0, "16-byte aligned arrays");
  double *const X = pool + VP::O_X, *const xs = pool + VP::O_XS, *const wt = pool + VP::O_WT, *const tab = pool + VP::O_TAB;
  const int wave_id = TPAR ? bx : b.wave_order[bx];
  const WaveMeta wv = b.wave[wave_id];
  SolverState &st = b.st[wv.win];
  if (lin_skip(st, mode)) return;
  const WinMeta wm = b.win[wv.win];
  const int lane = threadIdx.x;
  int cs[4], cn[4], ckm[4], cgo[4];
  const LaneSeg ls = lane_segment(wv, b.chunk, lane, cs, cn, ckm, cgo);
  const bool active = ls.active;
  const int n = wv.n_lanes, L = wm.L, s = ls.s;
  const bool prof = !TPAR && (wave_id == wm.wave_off) && lane == 0;
  long long c_proj = 0, c_gram = 0, c_t0 = pclk64(), c_a = 0;
  const double *xg = (mode ? b.xc : b.x) + (size_t)wv.win * XSTRIDE;
  double *lm_g_out = lin_lm_g(b, st, mode);
  double *wbase = b.lm_w + 80 * (size_t)wm.lm_off;
  const int li = ls.li;
  for (int e = lane; e < XSTRIDE; e += 64) xs[e] = xg[e];

  // Gram of a (start frame, t) slot on the FP64 matrix cores: X^T X with X = the 2 n corrected Jacobian rows (23 columns,
  // padded to 32) as three 16 x 16 tiles (0,0), (0,1), (1,1). One k-step = 4 rows = 2 landmarks; lane (lr, lk) supplies
  // X[row 4 kk + lk][lr] (tile column 0) and X[..][16 + lr] (tile column 1), which serve as A and B operands alike.
  // Every segment of the wave (its own start frame) accumulates into its own three tiles.
  // The first tile column holds everything a left-camera factor touches except td (factors.hpp: GC_*): while td is a constant block of
  // the solve (estimate_td: 0, estimator.cpp:1104) its rows need tile (0,0) only — one MFMA per k-step instead of three. (The
  // marginalisation keeps td: mode 0 always runs the full form.)
  const int lr = lane & 15, lk = lane >> 4;
  const int xoff = (lk >> 1) * XL + (lk & 1) * XR + lr;
  const bool c1on = lr < 7;    // columns 23 .. 31 of the second tile column do not exist
  const bool lean = COMPACT || (mode != 0 && (wm.const_mask & CONST_TD));
  // (coupling rows w: every row of a landmark's column is written exactly once — the observed poses and the extrinsic / td rows with their
  // sums, the rest with zeros at the end; TPAR: the workgroup of frame offset t owns the rows of pose s + t, k_visual_reduce the rest)
  for (int e = lane; e < 4 * XL + 16; e += 64) X[64 * XL + e] = 0.0;

  const double *obs = b.obs + wv.obs_off;
  const unsigned char *flg = b.flags + wv.flag_off;
  double oi[6];   // pts_i (3), vel_i (2), td_i: the observation in the start frame
  double lam = 1.0;
  for (int c = 0; c < 6; ++c) oi[c] = 0.0;
  oi[2] = 1.0;
  if (active) {
    lam = (mode ? b.lamc : b.lam)[ls.gi];
    oi[0] = obs[(size_t)0 * n + lane]; oi[1] = obs[(size_t)1 * n + lane]; oi[2] = obs[(size_t)2 * n + lane];
    oi[3] = obs[(size_t)6 * n + lane]; oi[4] = obs[(size_t)7 * n + lane]; oi[5] = obs[(size_t)10 * n + lane];
  }
  lds_barrier();
  // window-level table: the two extrinsic rotations as matrices, their translations, ric2^T ric
  if (lane < 9) {
    const m3 ric = qR(ldq_pose(xs + XO_EX)), ric2 = qR(ldq_pose(xs + XO_EX + 7));
    const m3 A2 = tr(ric2) * ric;
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;
#pragma unroll
    for (int q = 0; q < 9; ++q)
      if (q == lane) { e0 = ric.a[q]; e1 = ric2.a[q]; e2 = A2.a[q]; }
    wt[VW_RIC + lane] = e0; wt[VW_RIC2 + lane] = e1; wt[VW_A2 + lane] = e2;
    if (lane < 3) { wt[VW_TIC + lane] = xs[XO_EX + lane]; wt[VW_TIC2 + lane] = xs[XO_EX + 7 + lane]; }
  }
  lds_barrier();
  const double td = xs[XO_TD];
  VisLane VL;
  {
    const double dti = td - oi[5];
    VL.inv_lam = 1.0 / lam;
    VL.vix = oi[3]; VL.viy = oi[4];
    VL.pci = mk3((oi[0] - oi[3] * dti) * VL.inv_lam, (oi[1] - oi[4] * dti) * VL.inv_lam, oi[2] * VL.inv_lam);
    const double *ric = wt + VW_RIC, *tic = wt + VW_TIC;
    VL.p_i = mk3(ric[0] * VL.pci.x + ric[1] * VL.pci.y + ric[2] * VL.pci.z + tic[0], ric[3] * VL.pci.x + ric[4] * VL.pci.y + ric[5] * VL.pci.z + tic[1],
                 ric[6] * VL.pci.x + ric[7] * VL.pci.y + ric[8] * VL.pci.z + tic[2]);
    const double *pose_s = xs + XO_POSE + 7 * s;
    VL.p_w = qrot(ldq_pose(pose_s), VL.p_i) + ld3(pose_s);
  }
  const v3 pts_i = mk3(oi[0], oi[1], oi[2]);

  double E = 0.0, gl = 0.0, cost = 0.0;
  double wc_s[6], wc_e0[6], wc_e1[6], wc_td = 0.0;
  for (int c = 0; c < 6; ++c) wc_s[c] = wc_e0[c] = wc_e1[c] = 0.0;

  // observations of frame t are fetched one iteration ahead (11 coalesced loads in flight behind the previous frame's
  // factor evaluation and Gram pass; the barriers below are LDS-only and do not drain them)
  double on[11];
  unsigned char fl_next = 0;
  for (int c = 0; c < 11; ++c) on[c] = 0.0;
  if (active) {
    fl_next = flg[lane];
#pragma unroll
    for (int c = 0; c < 11; ++c) on[c] = obs[(size_t)c * n + lane];
  }
  const int t_begin = TPAR ? by : 0, t_end = TPAR ? min(by + 1, wv.kmax) : wv.kmax;
  if (TPAR && t_begin > 0 && active && t_begin < wv.kmax) {   // this workgroup's frame instead of frame 0
    fl_next = flg[(size_t)t_begin * n + lane];
    const double *obn = obs + (size_t)t_begin * 11 * n;
#pragma unroll
    for (int c = 0; c < 11; ++c) on[c] = obn[(size_t)c * n + lane];
  }
  // table-builder role of this lane: (frame of the pair, segment, camera, row)
  const int tb_tt = lane / 24, tb_g = (lane % 24) / 6, tb_kind = ((lane % 6) >= 3) ? 1 : 0, tb_r = lane % 3;
  int tb_s = 0, tb_km = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    if (g == tb_g) { tb_s = cs[g]; tb_km = (g < wv.nseg) ? ckm[g] : 0; }
  const int tb_base = max(t_begin, 1);   // (frame 0 is the one-frame factor: no pair)
  for (int t = t_begin; t < t_end; ++t) {
    const int tslot = (t >= tb_base) ? (t - tb_base) % VT_TB : 0;
    if (t >= tb_base && tslot == 0) {
      if (lane < 24 * VT_TB && t + tb_tt < tb_km && t + tb_tt < t_end)
        vis_build_pair_row(xs, wt, tb_s, min(tb_s + t + tb_tt, VILO_MAX_FRAMES - 1), tb_kind, tb_r, tab + (tb_tt * 4 + tb_g) * VT_N);
      lds_barrier();
    }
    const int j = min(s + t, VILO_MAX_FRAMES - 1);
    const unsigned char fl = fl_next;
    double ob[11];
#pragma unroll
    for (int c = 0; c < 11; ++c) ob[c] = on[c];
    if (!TPAR && active && t + 1 < wv.kmax) {
      fl_next = flg[(size_t)(t + 1) * n + lane];
      const double *obn = obs + (size_t)(t + 1) * 11 * n;
#pragma unroll
      for (int c = 0; c < 11; ++c) on[c] = obn[(size_t)c * n + lane];
    }
    // (COMPACT: G00 = the left camera's tile C0, G01 = the right camera's C1, G11 unused)
    mfma_d4 G00[4], G01[4], G11[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { G00[g] = mfma_d4{0.0, 0.0, 0.0, 0.0}; G01[g] = G00[g]; G11[g] = G00[g]; }
    double wj[6];
    for (int c = 0; c < 6; ++c) wj[c] = 0.0;
    const double *tb = tab + (tslot * 4 + max(ls.seg, 0)) * VT_N;
    v3 p_j = VL.p_i;
    if (t > 0) {
      const v3 d = mk3(VL.p_w.x - tb[VT_PJ], VL.p_w.y - tb[VT_PJ + 1], VL.p_w.z - tb[VT_PJ + 2]);
      p_j = mk3(tb[0] * d.x + tb[3] * d.y + tb[6] * d.z, tb[1] * d.x + tb[4] * d.y + tb[7] * d.z, tb[2] * d.x + tb[5] * d.y + tb[8] * d.z);
    }
    const double dtj = td - ob[10];

    for (int cam = (t == 0 ? 1 : 0); cam < 2; ++cam) {
      // cam 0: left observation (TwoFrameOneCam); cam 1: right observation (TwoFrameTwoCam, or OneFrameTwoCam at t == 0)
      const bool produce = active && (fl & 1) && (cam == 0 || (fl & 2));
      double *xr0 = &X[lane * XL], *xr1 = xr0 + XR;
      c_a = pclk64();
      if (produce) {
        double x0[XR], x1[XR], Jl[2], obc[5];
        double term[LM_NTERM], wjr[3];
        double rho0;
        if (COMPACT) {
          double tc[4][3];
          if (cam == 0) {
            obc[0] = ob[0]; obc[1] = ob[1]; obc[2] = ob[2]; obc[3] = ob[6]; obc[4] = ob[7];
            rho0 = vis_two_frame_c<0>(wt, tb, VL, p_j, obc, dtj, sq, huber_a, x0, x1, Jl, tc);
          } else {
            obc[0] = ob[3]; obc[1] = ob[4]; obc[2] = ob[5]; obc[3] = ob[8]; obc[4] = ob[9];
            if (t > 0) rho0 = vis_two_frame_c<1>(wt, tb, VL, p_j, obc, dtj, sq, huber_a, x0, x1, Jl, tc);
            else rho0 = vis_one_frame_c(wt, VL, pts_i, obc, dtj, sq, huber_a, x0, x1, Jl, tc);
          }
          term[0] = dot2(Jl[0], Jl[0], Jl[1], Jl[1]);
          term[1] = dot2(Jl[0], x0[GK_R], Jl[1], x1[GK_R]);
          const double pw = (t > 0) ? 1.0 : 0.0;   // the one-frame factor has no pose blocks (its B columns carry the tic column)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            term[2 + c] = pw * (dot2(x0[GK_B + c], Jl[0], x1[GK_B + c], Jl[1]));
            term[5 + c] = dot2(x0[GK_RI + c], Jl[0], x1[GK_RI + c], Jl[1]);
            term[8 + c] = dot2(tc[0][c], Jl[0], tc[1][c], Jl[1]);
            term[11 + c] = dot2(x0[GK_C0 + c], Jl[0], x1[GK_C0 + c], Jl[1]);
            term[14 + c] = dot2(tc[2][c], Jl[0], tc[3][c], Jl[1]);
            term[17 + c] = dot2(x0[GK_C1 + c], Jl[0], x1[GK_C1 + c], Jl[1]);
            wjr[c] = dot2(x0[GK_RJ + c], Jl[0], x1[GK_RJ + c], Jl[1]);
          }
          term[20] = 0.0;   // td is a constant block in this mode
        } else {
          x0[XR - 1] = 0.0; x1[XR - 1] = 0.0;
          if (cam == 0) {
            obc[0] = ob[0]; obc[1] = ob[1]; obc[2] = ob[2]; obc[3] = ob[6]; obc[4] = ob[7];
            rho0 = vis_two_frame<0>(wt, tb, VL, p_j, obc, dtj, sq, huber_a, x0, x1, Jl);
          } else {
            obc[0] = ob[3]; obc[1] = ob[4]; obc[2] = ob[5]; obc[3] = ob[8]; obc[4] = ob[9];
            if (t > 0) rho0 = vis_two_frame<1>(wt, tb, VL, p_j, obc, dtj, sq, huber_a, x0, x1, Jl);
            else rho0 = vis_one_frame(wt, VL, pts_i, obc, dtj, sq, huber_a, x0, x1, Jl);
          }
          // landmark-side reductions (the e-block of Ceres' Schur eliminator): the same 21 terms in both forms
          term[0] = dot2(Jl[0], Jl[0], Jl[1], Jl[1]);
          term[1] = dot2(Jl[0], x0[GC_R], Jl[1], x1[GC_R]);
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            term[2 + c] = dot2(x0[c], Jl[0], x1[c], Jl[1]);
            term[8 + c] = dot2(x0[GC_E0 + c], Jl[0], x1[GC_E0 + c], Jl[1]);
            term[14 + c] = dot2(x0[GC_E1 + c], Jl[0], x1[GC_E1 + c], Jl[1]);
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) wjr[c] = dot2(x0[GC_RJ + c], Jl[0], x1[GC_RJ + c], Jl[1]);
          term[20] = dot2(x0[GC_TD], Jl[0], x1[GC_TD], Jl[1]);
        }
        cost += rho0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          wj[c] -= term[2 + c];   // d r / d P_j = -d r / d P_i
          wj[3 + c] += wjr[c];
        }
        if (TPAR) {
          double *pt = b.lm_part + ((size_t)(t * 2 + cam) * LM_NTERM) * b.n_lm + ls.gi;
#pragma unroll
          for (int v = 0; v < LM_NTERM; ++v) pt[(size_t)v * b.n_lm] = term[v];
        } else {
          E += term[0];
          gl += term[1];
#pragma unroll
          for (int c = 0; c < 6; ++c) { wc_s[c] += term[2 + c]; wc_e0[c] += term[8 + c]; wc_e1[c] += term[14 + c]; }
          wc_td += term[20];
        }
#pragma unroll
        for (int c = 0; c < XR; ++c) { xr0[c] = x0[c]; xr1[c] = x1[c]; }
      } else {
#pragma unroll
        for (int c = 0; c < XR; ++c) { xr0[c] = 0.0; xr1[c] = 0.0; }
        if (TPAR && active) {
          // (a factor that does not exist contributes + 0.0 to the landmark's sums: written, so that nobody has to clear the terms first)
          double *pt = b.lm_part + ((size_t)(t * 2 + cam) * LM_NTERM) * b.n_lm + ls.gi;
#pragma unroll
          for (int v = 0; v < LM_NTERM; ++v) pt[(size_t)v * b.n_lm] = 0.0;
        }
      }
      lds_barrier();
      { const long long c_b = pclk64(); c_proj += c_b - c_a; c_a = c_b; }
      // rows of padding / unobserved lanes are zero, and every segment spans a multiple of 8 lanes = 4 k-steps:
      // the operands of the next trip are in flight behind the 12 MFMAs of this one
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g >= wv.nseg || t >= ckm[g]) continue;
        const int k0 = wv.seg_lane0[g] >> 1, k1 = k0 + (((cn[g] + 7) & ~7) >> 1);
        // Lanes lr >= 7 of the second tile column read past column 22 (the next row / lane: in bounds, arbitrary values): they only
        // reach rows / columns 23 .. 31 of the tiles, which nobody stores.
        double a0[4], a1[4], n0[4], n1[4];
        auto ldtrip = [&](int kk, double *p0, double *p1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double *xr = &X[2 * (kk + u) * XL + xoff];
            p0[u] = xr[0];
            p1[u] = xr[16];
          }
        };
        auto ldtrip0 = [&](int kk, double *p0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) p0[u] = X[2 * (kk + u) * XL + xoff];
        };
        auto dotrip = [&](const double *p0, const double *p1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            G00[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(p0[u], p0[u], G00[g], 0, 0, 0);
            G01[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(p0[u], p1[u], G01[g], 0, 0, 0);
            G11[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(p1[u], p1[u], G11[g], 0, 0, 0);
          }
        };
        mfma_d4 &G1t = (COMPACT && cam == 1) ? G01[g] : G00[g];   // the one tile of a lean trip
        auto dotrip0 = [&](const double *p0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) G1t = __builtin_amdgcn_mfma_f64_16x16x4f64(p0[u], p0[u], G1t, 0, 0, 0);
        };
        // two trips per turn, the operands of the next trip in flight behind the MFMAs of this one (a segment has an even number of
        // trips or is the last of its wave: the trip after its last one reads rows that exist — zero pad lanes at the end — and is dropped)
        // (sched_barrier: the scheduler otherwise sinks every load to just before its MFMA to save registers, and the wave waits out
        // an LDS round trip per k-step)
        if (COMPACT || (lean && cam == 0)) {
          ldtrip0(k0, a0);
          for (int kk0 = k0; kk0 < k1; kk0 += 8) {
            ldtrip0(min(kk0 + 4, 30), n0);
            __builtin_amdgcn_sched_barrier(0);
            dotrip0(a0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk0 + 4 < k1) {
              ldtrip0(min(kk0 + 8, 30), a0);
              __builtin_amdgcn_sched_barrier(0);
              dotrip0(n0);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        } else {
          ldtrip(k0, a0, a1);
          for (int kk0 = k0; kk0 < k1; kk0 += 8) {
            ldtrip(min(kk0 + 4, 30), n0, n1);
            __builtin_amdgcn_sched_barrier(0);
            dotrip(a0, a1);
            __builtin_amdgcn_sched_barrier(0);
            if (kk0 + 4 < k1) {
              ldtrip(min(kk0 + 8, 30), a0, a1);
              __builtin_amdgcn_sched_barrier(0);
              dotrip(n0, n1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      lds_barrier();
      c_gram += pclk64() - c_a;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g >= wv.nseg || t >= ckm[g]) continue;
      if (COMPACT) {
        // upper triangle of C0 + C1, then rows 0 .. 2 (the B rows) of C1: register 0 of the lanes lk = 0 .. 2
        double *gs = b.gram + (size_t)(cgo[g] + t) * VILO_GRAMC;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = lk + 4 * r;
          if (row <= lr) gs[tri16(row, lr)] = G00[g][r] + G01[g][r];
        }
        if (lk < 3) gs[VILO_GRAMC_TRI + 16 * lk + lr] = G01[g][0];
        continue;
      }
      double *gs = b.gram + (size_t)(cgo[g] + t) * VILO_GRAM;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lk + 4 * r;
        if (row <= lr) gs[tri23(row, lr)] = G00[g][r];
        if (c1on) gs[tri23(row, 16 + lr)] = G01[g][r];
        if (c1on && row < 7 && row <= lr) gs[tri23(16 + row, 16 + lr)] = G11[g][r];
      }
    }
    if (active && t > 0 && s + t < VILO_MAX_FRAMES) {
      const bool seen = fl & 1;
      for (int c = 0; c < 6; ++c) wbase[(size_t)(6 * j + c) * L + li] = seen ? wj[c] : 0.0;
    }
  }
  if (!TPAR && active) {
    // (rows of poses before the landmark's start frame are zero since vilo_batch_create and nobody writes them: not stored again)
    for (int f = s + wv.kmax; f < VILO_MAX_FRAMES; ++f)
      for (int c = 0; c < 6; ++c) wbase[(size_t)(6 * f + c) * L + li] = 0.0;
    wbase[(size_t)79 * L + li] = 0.0;
    b.lm_E[ls.gi] = E;
    lm_g_out[ls.gi] = gl;
    // (solve passes: the coupling rows of constant blocks are written as zeros, so the solvers need no masks in their Schur loops)
    const bool ex_on = mode == 0 || !(wm.const_mask & CONST_EX), td_on = mode == 0 || !(wm.const_mask & CONST_TD);
    for (int c = 0; c < 6; ++c) {
      wbase[(size_t)(6 * s + c) * L + li] = wc_s[c];
      wbase[(size_t)(CD_EX0 + c) * L + li] = ex_on ? wc_e0[c] : 0.0;
      wbase[(size_t)(CD_EX1 + c) * L + li] = ex_on ? wc_e1[c] : 0.0;
    }
    wbase[(size_t)CD_TD * L + li] = td_on ? wc_td : 0.0;
  }
  // robust cost of the evaluated point: per packed wave (walking form: slot 0 of the wave's frame slots, the rest zero) or per (packed
  // wave, frame); k_accept adds the slots of a window in a fixed order
  {
    const double csum = wave_sum(active ? cost : 0.0);
    double *cost_out = b.chunk_cost + (size_t)wave_id * VILO_MAX_FRAMES;
    if (TPAR) { if (lane == 0) cost_out[by] = csum; }
    else if (lane < VILO_MAX_FRAMES) cost_out[lane] = (lane == 0) ? csum : 0.0;
  }
  PCLK(if (prof) { st.phase_clk[28] = clock64() - c_t0; st.phase_clk[29] = c_proj; st.phase_clk[30] = c_gram; st.phase_clk[31] = wv.n_lanes; st.phase_clk[32] = wv.kmax; });
}

__global__ void __launch_bounds__(64) k_visual_linearize(BatchDev b, double sq, double huber_a, int mode) {
  __shared__ __attribute__((aligned(16))) double pool[VisPool<false>::N];
  visual_linearize_body<false, false>(b, sq, huber_a, mode, pool, blockIdx.x, 0);
}
__global__ void __launch_bounds__(64) k_visual_linearize_tpar(BatchDev b, double sq, double huber_a, int mode) {
  __shared__ __attribute__((aligned(16))) double pool[VisPool<false>::N];
  visual_linearize_body<true, false>(b, sq, huber_a, mode, pool, blockIdx.x, blockIdx.y);
}
// the compact forms (solve passes, td constant in every window of the batch)
__global__ void __launch_bounds__(64) k_visual_linearize_c(BatchDev b, double sq, double huber_a, int mode) {
  __shared__ __attribute__((aligned(16))) double pool[VisPool<true>::N];
  visual_linearize_body<false, true>(b, sq, huber_a, mode, pool, blockIdx.x, 0);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_visual_linearize_tpar_c(BatchDev b, double sq, double huber_a, int mode) {
  __shared__ __attribute__((aligned(16))) double pool[VisPool<true>::N];
  visual_linearize_body<true, true>(b, sq, huber_a, mode, pool, blockIdx.x, blockIdx.y);
}

// =================================================================================================
// k_visual_linearize_pc: the compact walking form as a PRODUCER / CONSUMER pair of waves
// =================================================================================================
// One workgroup of two waves per packed wave. The single-wave form alternates two phases that use different parts of the SIMD and wait
// for each other: evaluating the factors of a (frame, camera) step into LDS rows (FP64 VALU + table reads, latency-bound: ~3.8 k
// cycles per step for ~1.2 k cycles of issue) and their Gram on the matrix cores (~2.8 k). Here wave P evaluates step n + 1 into one
// half of a double-buffered row image while wave C runs the MFMA pass of step n out of the other half, one s_barrier per step:
//     P: eval(0) | eval(1)  | eval(2)  | ...          C:         | mfma(0)  | mfma(1)  | ...
// P keeps what the landmark side needs (sums, coupling rows, cost) and no accumulators, C keeps the eight Gram tiles and nothing else:
// both roles fit 256 registers, so the two waves of a workgroup share the chip with the two of another (eight waves per CU as
// before, but every packed wave advances at the pace of its slower role instead of the sum of both). Same arithmetic per factor and
// per Gram tile as k_visual_linearize_c: bitwise the same slots, landmark sums and coupling rows.
#define PC_XN (64 * XLANEC)   // rows of one buffer: 64 lanes x (2 rows x 16 + 2 pad); a trip past the last lane is clamped, not padded
#define PC_POOL_N (2 * PC_XN + XSTRIDE + 40 + 4 * VT_N)
__device__ __forceinline__ void imu_fused_body(BatchDev &b, int f, double g_norm, int mode, double *lds);
#define IMU_FUSED_LDS 1800   // doubles of one wave of imu_fused_body (below)
static_assert(2 * IMU_FUSED_LDS <= PC_POOL_N, "two IMU waves fit the visual pair's pool");
// WITH_IMU (k_visual_linearize_pc_imu; small batches, vilo_solve_launch): workgroups behind the packed waves' linearise the IMU(-leg)
// factors, one per wave (imu_fused_body): the IMU pass of the iteration runs under the visual one instead of after it. The full-batch
// kernel is the instantiation without them (its register allocation is not to be disturbed by the other role's).
template <bool WITH_IMU>
__device__ __forceinline__ void visual_linearize_pc_body(BatchDev &b, double sq, double huber_a, int mode, double gn, double *pool, int imu_first) {
  // WITH_IMU: the first (W * 10 + 1) / 2 workgroups are the IMU factors', one factor per wave. First, because they are the short ones (a
  // third of the longest packed wave's time): the packed waves are launched longest first, so what has to wait for a slot behind the IMU
  // workgroups are the short packed waves, which still end before the long ones. (Behind the packed waves the IMU workgroups of 128
  // windows start when the first packed waves retire and end last: 57 instead of 49 us; two factors per wave to make them all resident
  // from the start: slower still, 128 windows 545 -> 536 k, 256 windows 963 -> 863 k window-iterations/s.)
  const int n_imu_wg = WITH_IMU ? (b.W * 10 + 1) / 2 : 0;
  const int imu_lo = imu_first ? 0 : b.n_waves;
  if (WITH_IMU && (int)blockIdx.x >= imu_lo && (int)blockIdx.x < imu_lo + n_imu_wg) {
    const int wv_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), f = 2 * ((int)blockIdx.x - imu_lo) + wv_;
    if (f < b.W * 10) imu_fused_body(b, f, gn, mode, pool + wv_ * IMU_FUSED_LDS);
    return;
  }
  double *const X = pool, *const xs = pool + 2 * PC_XN, *const wt = xs + XSTRIDE, *const tab = wt + 40;   // tab: the pair tables of ONE frame (four segments)
  const int wave_id = b.wave_order[(int)blockIdx.x - (imu_first ? n_imu_wg : 0)];
  const WaveMeta wv = b.wave[wave_id];
  SolverState &st = b.st[wv.win];
  if (lin_skip(st, mode)) return;
  const WinMeta wm = b.win[wv.win];
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0: producer, 1: consumer
  const int lane = threadIdx.x & 63;
  int cs[4], cn[4], ckm[4], cgo[4];
  const LaneSeg ls = lane_segment(wv, b.chunk, lane, cs, cn, ckm, cgo);
  const int kmax = wv.kmax;
  auto step_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  if (role == 1) {
    // ------------------------------------------ consumer: Gram tiles of every (segment, frame) slot ------------------------------------------
    const int lr = lane & 15, lk = lane >> 4;
    const int xoff = (lk >> 1) * XLANEC + (lk & 1) * XROWC + lr;
    mfma_d4 G0[4], G1[4];   // left / right camera tile of the frame in progress, per segment
#pragma unroll
    for (int g = 0; g < 4; ++g) { G0[g] = mfma_d4{0.0, 0.0, 0.0, 0.0}; G1[g] = G0[g]; }
    // one (frame, camera) step: the MFMA pass over every live segment's rows (CAM is a compile-time constant: the tiles stay in registers)
    auto pass = [&](int t, auto cam_c) {
      constexpr int CAM = decltype(cam_c)::value;
      const int nstep = (t == 0) ? 0 : 2 * t - 1 + CAM;
      const double *Xb = X + (nstep & 1) * PC_XN;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g >= wv.nseg || t >= ckm[g]) continue;
        const int k0 = wv.seg_lane0[g] >> 1, k1 = k0 + (((cn[g] + 7) & ~7) >> 1);
        mfma_d4 G = CAM ? G1[g] : G0[g];
        double a0[4], n0[4];
        auto ldtrip0 = [&](int kk, double *p0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) p0[u] = Xb[2 * (kk + u) * XLANEC + xoff];
        };
        auto dotrip0 = [&](const double *p0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) G = __builtin_amdgcn_mfma_f64_16x16x4f64(p0[u], p0[u], G, 0, 0, 0);
        };
        // two trips per turn, the operands of the next trip in flight behind the MFMAs of this one (a trip past the segment's last
        // one is loaded from rows that exist — clamped to the last trip of the image — and dropped)
        ldtrip0(k0, a0);
        for (int kk0 = k0; kk0 < k1; kk0 += 8) {
          ldtrip0(min(kk0 + 4, 28), n0);
          __builtin_amdgcn_sched_barrier(0);
          dotrip0(a0);
          __builtin_amdgcn_sched_barrier(0);
          if (kk0 + 4 < k1) {
            ldtrip0(min(kk0 + 8, 28), a0);
            __builtin_amdgcn_sched_barrier(0);
            dotrip0(n0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (CAM) G1[g] = G; else G0[g] = G;
      }
    };
    // The landmark side of the step (lane = landmark, as in the producer): the producer is the pair's critical path (it waits 2 % of its
    // time at the step barriers, this wave 50 %), so the reductions that need nothing but a lane's two rows and its 2 x 1 landmark
    // Jacobian (handed over in the two pad doubles of the lane's row block) are formed here — the same 15 + 2 terms, formed and added
    // exactly as in the single-wave / frame-parallel forms (bitwise); the six terms of the extrinsic translations stay with the producer.
    const bool active = ls.active;
    const int s_lm = ls.s, li = ls.li, L = wm.L;
    double *wbase = b.lm_w + 80 * (size_t)wm.lm_off;
    double E = 0.0, gl = 0.0, wc_s[6], wc_e0h[3], wc_e1h[3], wj[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) wc_s[c] = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { wc_e0h[c] = 0.0; wc_e1h[c] = 0.0; }
    auto reduce = [&](int t, int cam) {
      const int nstep = (t == 0) ? 0 : 2 * t - 1 + cam;
      const double *xr0 = X + (nstep & 1) * PC_XN + lane * XLANEC, *xr1 = xr0 + XROWC;
      double x0[XROWC], x1[XROWC], Jl[2];
#pragma unroll
      for (int c = 0; c < XROWC; ++c) { x0[c] = xr0[c]; x1[c] = xr1[c]; }
      Jl[0] = xr0[2 * XROWC]; Jl[1] = xr0[2 * XROWC + 1];
      double term[LM_NTERM], wjr[3];
      term[0] = dot2(Jl[0], Jl[0], Jl[1], Jl[1]);
      term[1] = dot2(Jl[0], x0[GK_R], Jl[1], x1[GK_R]);
      const double pw = (t > 0) ? 1.0 : 0.0;   // the one-frame factor has no pose blocks (its B columns carry the tic column)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        term[2 + c] = pw * (dot2(x0[GK_B + c], Jl[0], x1[GK_B + c], Jl[1]));
        term[5 + c] = dot2(x0[GK_RI + c], Jl[0], x1[GK_RI + c], Jl[1]);
        term[11 + c] = dot2(x0[GK_C0 + c], Jl[0], x1[GK_C0 + c], Jl[1]);
        term[17 + c] = dot2(x0[GK_C1 + c], Jl[0], x1[GK_C1 + c], Jl[1]);
        wjr[c] = dot2(x0[GK_RJ + c], Jl[0], x1[GK_RJ + c], Jl[1]);
        term[8 + c] = 0.0; term[14 + c] = 0.0;   // (the producer's)
      }
      term[20] = 0.0;
      // (every term is a value of its own before it is added — the frame-parallel form stores it, the sums must round alike: opaque to the
      // optimiser so that no multiply of a term is contracted into the accumulation)
#pragma unroll
      for (int v = 0; v < 20; ++v) asm volatile("" : "+v"(term[v]));
#pragma unroll
      for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(wjr[c]));
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        wj[c] -= term[2 + c];   // d r / d P_j = -d r / d P_i
        wj[3 + c] += wjr[c];
      }
      E += term[0];
      gl += term[1];
#pragma unroll
      for (int c = 0; c < 6; ++c) wc_s[c] += term[2 + c];
#pragma unroll
      for (int c = 0; c < 3; ++c) { wc_e0h[c] += term[11 + c]; wc_e1h[c] += term[17 + c]; }
    };
    long long c_wait = 0, cw0 = 0;   // (profiling build only: cycles this wave spends at the step barriers)
    const long long c_start = pclk64();
    step_barrier();   // step 0's rows are there
    for (int t = 0; t < kmax; ++t) {
#pragma unroll
      for (int c = 0; c < 6; ++c) wj[c] = 0.0;
      if (t > 0) {
        pass(t, std::integral_constant<int, 0>{});
        reduce(t, 0);
        PCLK(cw0 = clock64());
        step_barrier();
        PCLK(c_wait += clock64() - cw0);
      }
      pass(t, std::integral_constant<int, 1>{});
      reduce(t, 1);
      if (active && t > 0 && s_lm + t < VILO_MAX_FRAMES) {   // (a frame that does not see the landmark produced zero rows: wj is zero)
        const int j = s_lm + t;
#pragma unroll
        for (int c = 0; c < 6; ++c) wbase[(size_t)(6 * j + c) * L + li] = wj[c];
      }
      // the frame's slots: upper triangle of C0 + C1, then rows 0 .. 2 (the B rows) of C1: register 0 of the lanes lk = 0 .. 2
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g >= wv.nseg || t >= ckm[g]) continue;
        double *gs = b.gram + (size_t)(cgo[g] + t) * VILO_GRAMC;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = lk + 4 * r;
          if (row <= lr) gs[tri16(row, lr)] = G0[g][r] + G1[g][r];
        }
        if (lk < 3) gs[VILO_GRAMC_TRI + 16 * lk + lr] = G1[g][0];
        G0[g] = mfma_d4{0.0, 0.0, 0.0, 0.0}; G1[g] = G0[g];
      }
      PCLK(cw0 = clock64());
      step_barrier();
      PCLK(c_wait += clock64() - cw0);
    }
    PCLK(if (wave_id == 0 && lane == 0) { st.phase_clk[28] = clock64() - c_start; st.phase_clk[29] = c_wait; st.phase_clk[32] = kmax; });
    if (active) {
      b.lm_E[ls.gi] = E;
      lin_lm_g(b, st, mode)[ls.gi] = gl;
      const bool ex_on = mode == 0 || !(wm.const_mask & CONST_EX);
#pragma unroll
      for (int c = 0; c < 6; ++c) wbase[(size_t)(6 * s_lm + c) * L + li] = wc_s[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        wbase[(size_t)(CD_EX0 + 3 + c) * L + li] = ex_on ? wc_e0h[c] : 0.0;
        wbase[(size_t)(CD_EX1 + 3 + c) * L + li] = ex_on ? wc_e1h[c] : 0.0;
      }
    }
    return;
  }

  // ------------------------------------------ producer: factor evaluation, landmark side ------------------------------------------
  const bool active = ls.active;
  const int n = wv.n_lanes, L = wm.L, s = ls.s;
  const double *xg = (mode ? b.xc : b.x) + (size_t)wv.win * XSTRIDE;
  double *wbase = b.lm_w + 80 * (size_t)wm.lm_off;
  const int li = ls.li;
  for (int e = lane; e < XSTRIDE; e += 64) xs[e] = xg[e];
  const double *obs = b.obs + wv.obs_off;
  const unsigned char *flg = b.flags + wv.flag_off;
  double oi[6];   // pts_i (3), vel_i (2), td_i: the observation in the start frame
  double lam = 1.0;
  for (int c = 0; c < 6; ++c) oi[c] = 0.0;
  oi[2] = 1.0;
  if (active) {
    lam = (mode ? b.lamc : b.lam)[ls.gi];
    oi[0] = obs[(size_t)0 * n + lane]; oi[1] = obs[(size_t)1 * n + lane]; oi[2] = obs[(size_t)2 * n + lane];
    oi[3] = obs[(size_t)6 * n + lane]; oi[4] = obs[(size_t)7 * n + lane]; oi[5] = obs[(size_t)10 * n + lane];
  }
  lds_fence();   // (xs, wt, tab belong to this wave alone: its own LDS traffic in order is all that is needed)
  if (lane < 9) {
    const m3 ric = qR(ldq_pose(xs + XO_EX)), ric2 = qR(ldq_pose(xs + XO_EX + 7));
    const m3 A2 = tr(ric2) * ric;
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;
#pragma unroll
    for (int q = 0; q < 9; ++q)
      if (q == lane) { e0 = ric.a[q]; e1 = ric2.a[q]; e2 = A2.a[q]; }
    wt[VW_RIC + lane] = e0; wt[VW_RIC2 + lane] = e1; wt[VW_A2 + lane] = e2;
    if (lane < 3) { wt[VW_TIC + lane] = xs[XO_EX + lane]; wt[VW_TIC2 + lane] = xs[XO_EX + 7 + lane]; }
  }
  lds_fence();
  const double td = xs[XO_TD];
  VisLane VL;
  {
    const double dti = td - oi[5];
    VL.inv_lam = 1.0 / lam;
    VL.vix = oi[3]; VL.viy = oi[4];
    VL.pci = mk3((oi[0] - oi[3] * dti) * VL.inv_lam, (oi[1] - oi[4] * dti) * VL.inv_lam, oi[2] * VL.inv_lam);
    const double *ric = wt + VW_RIC, *tic = wt + VW_TIC;
    VL.p_i = mk3(ric[0] * VL.pci.x + ric[1] * VL.pci.y + ric[2] * VL.pci.z + tic[0], ric[3] * VL.pci.x + ric[4] * VL.pci.y + ric[5] * VL.pci.z + tic[1],
                 ric[6] * VL.pci.x + ric[7] * VL.pci.y + ric[8] * VL.pci.z + tic[2]);
    const double *pose_s = xs + XO_POSE + 7 * s;
    VL.p_w = qrot(ldq_pose(pose_s), VL.p_i) + ld3(pose_s);
  }
  const v3 pts_i = mk3(oi[0], oi[1], oi[2]);
  double cost = 0.0;
  double wc_e0[3], wc_e1[3];   // the extrinsic translations' share of the landmark's coupling rows (the rest: the consumer)
  for (int c = 0; c < 3; ++c) wc_e0[c] = wc_e1[c] = 0.0;
  double on[11];
  unsigned char fl_next = 0;
  for (int c = 0; c < 11; ++c) on[c] = 0.0;
  if (active) {
    fl_next = flg[lane];
#pragma unroll
    for (int c = 0; c < 11; ++c) on[c] = obs[(size_t)c * n + lane];
  }
  // table-builder role of this lane: (segment, camera, row) = 4 x 2 x 3
  const int tb_g = lane / 6, tb_kind = ((lane % 6) >= 3) ? 1 : 0, tb_r = lane % 3;
  int tb_s = 0, tb_km = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    if (g == tb_g) { tb_s = cs[g]; tb_km = (g < wv.nseg) ? ckm[g] : 0; }
  long long p_wait = 0, pw0 = 0;   // (profiling build only)
  long long p_tab = 0, p_eval = 0, p_store = 0, pk0 = 0, pk1 = 0;
  const long long p_start = pclk64();
  for (int t = 0; t < kmax; ++t) {
    PCLK(pk0 = clock64());
    if (t >= 1) {
      if (lane < 24 && t < tb_km) vis_build_pair_row(xs, wt, tb_s, min(tb_s + t, VILO_MAX_FRAMES - 1), tb_kind, tb_r, tab + tb_g * VT_N);
      lds_fence();   // (a compiler barrier alone — a wave's LDS operations execute in issue order — measured the same: 0.679 ms either way)
    }
    PCLK(p_tab += clock64() - pk0);
    const int j = min(s + t, VILO_MAX_FRAMES - 1);
    const unsigned char fl = fl_next;
    double ob[11];
#pragma unroll
    for (int c = 0; c < 11; ++c) ob[c] = on[c];
    if (active && t + 1 < kmax) {
      fl_next = flg[(size_t)(t + 1) * n + lane];
      const double *obn = obs + (size_t)(t + 1) * 11 * n;
#pragma unroll
      for (int c = 0; c < 11; ++c) on[c] = obn[(size_t)c * n + lane];
    }
    const double *tb = tab + max(ls.seg, 0) * VT_N;
    v3 p_j = VL.p_i;
    if (t > 0) {
      const v3 d = mk3(VL.p_w.x - tb[VT_PJ], VL.p_w.y - tb[VT_PJ + 1], VL.p_w.z - tb[VT_PJ + 2]);
      p_j = mk3(tb[0] * d.x + tb[3] * d.y + tb[6] * d.z, tb[1] * d.x + tb[4] * d.y + tb[7] * d.z, tb[2] * d.x + tb[5] * d.y + tb[8] * d.z);
    }
    const double dtj = td - ob[10];
    for (int cam = (t == 0 ? 1 : 0); cam < 2; ++cam) {
      const int nstep = (t == 0) ? 0 : 2 * t - 1 + cam;
      const bool produce = active && (fl & 1) && (cam == 0 || (fl & 2));
      double *xr0 = X + (nstep & 1) * PC_XN + lane * XLANEC, *xr1 = xr0 + XROWC;
      PCLK(pk0 = clock64());
      if (produce) {
        double x0[XROWC], x1[XROWC], Jl[2], obc[5], tc[4][3];
        double rho0;
        if (cam == 0) {
          obc[0] = ob[0]; obc[1] = ob[1]; obc[2] = ob[2]; obc[3] = ob[6]; obc[4] = ob[7];
          rho0 = vis_two_frame_c<0>(wt, tb, VL, p_j, obc, dtj, sq, huber_a, x0, x1, Jl, tc);
        } else {
          obc[0] = ob[3]; obc[1] = ob[4]; obc[2] = ob[5]; obc[3] = ob[8]; obc[4] = ob[9];
          if (t > 0) rho0 = vis_two_frame_c<1>(wt, tb, VL, p_j, obc, dtj, sq, huber_a, x0, x1, Jl, tc);
          else rho0 = vis_one_frame_c(wt, VL, pts_i, obc, dtj, sq, huber_a, x0, x1, Jl, tc);
        }
        // landmark side: the six terms that need the explicit extrinsic-translation columns; the others are the consumer's
        double te0[3], te1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          te0[c] = dot2(tc[0][c], Jl[0], tc[1][c], Jl[1]);
          te1[c] = dot2(tc[2][c], Jl[0], tc[3][c], Jl[1]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { asm volatile("" : "+v"(te0[c])); asm volatile("" : "+v"(te1[c])); }
#pragma unroll
        for (int c = 0; c < 3; ++c) { wc_e0[c] += te0[c]; wc_e1[c] += te1[c]; }
        cost += rho0;
        PCLK(asm volatile("" ::: "memory"); pk1 = clock64(); p_eval += pk1 - pk0);
#pragma unroll
        for (int c = 0; c < XROWC; ++c) { xr0[c] = x0[c]; xr1[c] = x1[c]; }
        xr0[2 * XROWC] = Jl[0]; xr0[2 * XROWC + 1] = Jl[1];
      } else {
#pragma unroll
        for (int c = 0; c < XROWC; ++c) { xr0[c] = 0.0; xr1[c] = 0.0; }
        xr0[2 * XROWC] = 0.0; xr0[2 * XROWC + 1] = 0.0;
      }
      PCLK(pw0 = clock64(); if (produce) p_store += pw0 - pk1);
      step_barrier();
      PCLK(p_wait += clock64() - pw0);
    }
  }
  PCLK(if (wave_id == 0 && lane == 0) { st.phase_clk[30] = clock64() - p_start; st.phase_clk[31] = p_wait; st.phase_clk[59] = p_tab; st.phase_clk[60] = p_eval; st.phase_clk[61] = p_store; });
  if (active) {
    // (rows of poses before the landmark's start frame are zero since vilo_batch_create and nobody writes them: not stored again)
    for (int f = s + kmax; f < VILO_MAX_FRAMES; ++f)
      for (int c = 0; c < 6; ++c) wbase[(size_t)(6 * f + c) * L + li] = 0.0;
    wbase[(size_t)79 * L + li] = 0.0;
    const bool ex_on = mode == 0 || !(wm.const_mask & CONST_EX);
    for (int c = 0; c < 3; ++c) {
      wbase[(size_t)(CD_EX0 + c) * L + li] = ex_on ? wc_e0[c] : 0.0;
      wbase[(size_t)(CD_EX1 + c) * L + li] = ex_on ? wc_e1[c] : 0.0;
    }
    wbase[(size_t)CD_TD * L + li] = 0.0;   // (td is a constant block in the compact mode)
  }
  {
    const double csum = wave_sum(active ? cost : 0.0);
    double *cost_out = b.chunk_cost + (size_t)wave_id * VILO_MAX_FRAMES;
    if (lane < VILO_MAX_FRAMES) cost_out[lane] = (lane == 0) ? csum : 0.0;
  }
  step_barrier();   // (the consumer's last step)
}
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) k_visual_linearize_pc(BatchDev b, double sq, double huber_a, int mode) {
  __shared__ __attribute__((aligned(16))) double pool[PC_POOL_N];
  visual_linearize_pc_body<false>(b, sq, huber_a, mode, 0.0, pool, 0);
}
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) k_visual_linearize_pc_imu(BatchDev b, double sq, double huber_a, int mode, double gn, int imu_first) {
  __shared__ __attribute__((aligned(16))) double pool[PC_POOL_N];
  visual_linearize_pc_body<true>(b, sq, huber_a, mode, gn, pool, imu_first);
}

__global__ void __launch_bounds__(64) k_visual_reduce(BatchDev b, int mode) { visual_reduce_body(b, blockIdx.x, mode, false); }

// both forms behind one call (kernel kind 0 of the profiling table).
// fuse_imu (solve passes of small batches with compact rows; gn = g_norm): the IMU factors are linearised by extra workgroups of the same
// launch (imu_fused_body) and the second half of the frame-parallel form is left to k_assemble_s's extra workgroups (reduce_later).
__global__ void k_lin_small_c(BatchDev b, double sq, double huber_a, double gn, int mode, int n_imu);   // (below, behind the IMU kernels)
static void launch_visual_linearize(BatchDev &b, double sq, double ha, hipStream_t s, int mode, bool fuse_imu = false, double gn = 0.0) {
  if (b.n_waves <= 0) return;
  const bool compact = b.compact && mode != 0;   // (the marginalisation's pass keeps td: full 23-column slots)
  if (b.lm_part) {
    // (every (packed wave, frame) workgroup writes all the terms and coupling rows it owns, zeros included: nothing to clear first)
    if (compact && fuse_imu) {
      hipLaunchKernelGGL(k_lin_small_c, dim3(b.W * 10 + b.n_waves * VILO_MAX_FRAMES), dim3(64), 0, s, b, sq, ha, gn, mode, b.W * 10);
      return;   // (k_assemble_s's extra workgroups finish the frame-parallel form)
    }
    if (compact) hipLaunchKernelGGL(k_visual_linearize_tpar_c, dim3(b.n_waves, VILO_MAX_FRAMES), dim3(64), 0, s, b, sq, ha, mode);
    else hipLaunchKernelGGL(k_visual_linearize_tpar, dim3(b.n_waves, VILO_MAX_FRAMES), dim3(64), 0, s, b, sq, ha, mode);
    hipLaunchKernelGGL(k_visual_reduce, dim3(b.n_waves), dim3(64), 0, s, b, mode);
  } else {
    // (VILO_VISUAL_FORM=single keeps the one-wave compact form: A/B runs)
    static const bool pc = [] { const char *e = getenv("VILO_VISUAL_FORM"); return !(e && !strcmp(e, "single")); }();
    if (compact && pc) {
      // (order of the two kinds of workgroup, measured in the captured launch sequence: IMU first 564 k / last 430 k window-iterations/s at
      // 128 windows, 986 / 838 k at 256; 1018 / 1054 k at 384, 1227 / 1239 k at 512. VILO_IMU_FIRST = 0 / 1 pins it.)
      static const int imu_first_env = [] { const char *e = getenv("VILO_IMU_FIRST"); return e ? atoi(e) : -1; }();
      const int imu_first = imu_first_env >= 0 ? imu_first_env : (b.W <= 256 ? 1 : 0);
      if (fuse_imu) hipLaunchKernelGGL(k_visual_linearize_pc_imu, dim3(b.n_waves + (b.W * 10 + 1) / 2), dim3(128), 0, s, b, sq, ha, mode, gn, imu_first);
      else hipLaunchKernelGGL(k_visual_linearize_pc, dim3(b.n_waves), dim3(128), 0, s, b, sq, ha, mode);
    }
    else if (compact) hipLaunchKernelGGL(k_visual_linearize_c, dim3(b.n_waves), dim3(64), 0, s, b, sq, ha, mode);
    else hipLaunchKernelGGL(k_visual_linearize, dim3(b.n_waves), dim3(64), 0, s, b, sq, ha, mode);
  }
}
// does launch_visual_linearize(fuse_imu = true) take the IMU factors along for this batch?
static bool visual_launch_takes_imu(const BatchDev &b) {
  static const bool pc = [] { const char *e = getenv("VILO_VISUAL_FORM"); return !(e && !strcmp(e, "single")); }();
  return b.n_waves > 0 && b.compact && (b.lm_part || pc);
}

// Residual-only evaluation at the candidate point (TrustRegionMinimizer::ComputeCandidatePointAndEvaluateCost).
// (The candidate inverse depths lambda_c are formed by the solver.)
// One workgroup per (packed wave, frame offset t): 33 k short waves instead of 3 k waves walking up to 11 frames each — the pass
// is latency-bound, the frames of a landmark are independent here, and k_accept adds the per-(wave, t) partial sums.
__global__ void __launch_bounds__(64) k_visual_cost(BatchDev b, double sq, double huber_a) {
  const int wave_id = xcd_balanced(blockIdx.x, gridDim.x);
  const WaveMeta wv = b.wave[wave_id];
  const SolverState &st = b.st[wv.win];
  if (st.done) return;
  const int lane = threadIdx.x, t = blockIdx.y;
  double *cost_out = b.chunk_cost + (size_t)wave_id * VILO_MAX_FRAMES + t;
  if (t >= wv.kmax) { if (lane == 0) *cost_out = 0.0; return; }
  int cs[4], cn[4], ckm[4], cgo[4];
  const LaneSeg ls = lane_segment(wv, b.chunk, lane, cs, cn, ckm, cgo);
  const bool active = ls.active;
  const int n = wv.n_lanes, s = ls.s;
  const double *x = b.xc + (size_t)wv.win * XSTRIDE;
  const double *obs = b.obs + wv.obs_off;
  const unsigned char *flg = b.flags + wv.flag_off;
  double cost = 0.0;
  if (active) {
    const int gi = ls.gi;
    const double lam = b.lamc[gi];
    const unsigned char fl = flg[(size_t)t * n + lane];
    if (fl & 1) {
      double o12[12];
      o12[0] = obs[(size_t)0 * n + lane]; o12[1] = obs[(size_t)1 * n + lane]; o12[2] = obs[(size_t)2 * n + lane];
      o12[6] = obs[(size_t)6 * n + lane]; o12[7] = obs[(size_t)7 * n + lane]; o12[10] = obs[(size_t)10 * n + lane];
      const double *pose_s = x + XO_POSE + 7 * s, *ex0 = x + XO_EX, *ex1 = x + XO_EX + 7;
      const double td = x[XO_TD];
      const double *pose_j = x + XO_POSE + 7 * min(s + t, VILO_MAX_FRAMES - 1);
      const double *ob = obs + (size_t)t * 11 * n;
      o12[11] = ob[(size_t)10 * n + lane];
      double r[2];
      if (t > 0) {
        o12[3] = ob[(size_t)0 * n + lane]; o12[4] = ob[(size_t)1 * n + lane]; o12[5] = ob[(size_t)2 * n + lane];
        o12[8] = ob[(size_t)6 * n + lane]; o12[9] = ob[(size_t)7 * n + lane];
        proj_factor<0>(o12, pose_s, pose_j, ex0, ex1, lam, td, sq, r, false, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        double rho[3];
        huber_rho(huber_a, r[0] * r[0] + r[1] * r[1], rho);
        cost += rho[0];
      }
      if (fl & 2) {
        o12[3] = ob[(size_t)3 * n + lane]; o12[4] = ob[(size_t)4 * n + lane]; o12[5] = ob[(size_t)5 * n + lane];
        o12[8] = ob[(size_t)8 * n + lane]; o12[9] = ob[(size_t)9 * n + lane];
        if (t > 0) proj_factor<1>(o12, pose_s, pose_j, ex0, ex1, lam, td, sq, r, false, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        else proj_factor<2>(o12, pose_s, pose_j, ex0, ex1, lam, td, sq, r, false, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        double rho[3];
        huber_rho(huber_a, r[0] * r[0] + r[1] * r[1], rho);
        cost += rho[0];
      }
    }
  }
  const double csum = wave_sum(active ? cost : 0.0);
  if (lane == 0) *cost_out = csum;
}

// The same for large batches: one wave per packed wave walks its frames (lane = landmark). The start-frame observation, 1 / lambda and the
// world point are formed once per landmark and every observation is read once (the per-(wave, frame) form reads the start frame's
// with every frame: 2 x the bytes); the partial-cost slots keep the layout k_accept sums (slot 0 of the wave, the rest zero).
__global__ void __launch_bounds__(64) k_visual_cost_walk(BatchDev b, double sq, double huber_a) {
  const int wave_id = b.wave_order[blockIdx.x];
  const WaveMeta wv = b.wave[wave_id];
  const SolverState &st = b.st[wv.win];
  if (st.done) return;
  const int lane = threadIdx.x;
  int cs[4], cn[4], ckm[4], cgo[4];
  const LaneSeg ls = lane_segment(wv, b.chunk, lane, cs, cn, ckm, cgo);
  const int n = wv.n_lanes, s = ls.s;
  const double *x = b.xc + (size_t)wv.win * XSTRIDE;
  const double *obs = b.obs + wv.obs_off;
  const unsigned char *flg = b.flags + wv.flag_off;
  double cost = 0.0;
  if (ls.active) {
    const double td = x[XO_TD];
    const double *ex0 = x + XO_EX, *ex1 = x + XO_EX + 7;
    const quat qic = ldq_pose(ex0), qic2 = ldq_pose(ex1);
    const v3 tic = ld3(ex0), tic2 = ld3(ex1);
    const double inv_lam = 1.0 / b.lamc[ls.gi];
    const double dti = td - obs[(size_t)10 * n + lane];
    const v3 pci = mk3((obs[(size_t)0 * n + lane] - obs[(size_t)6 * n + lane] * dti) * inv_lam, (obs[(size_t)1 * n + lane] - obs[(size_t)7 * n + lane] * dti) * inv_lam,
                       obs[(size_t)2 * n + lane] * inv_lam);
    const v3 p_i = qrot(qic, pci) + tic;
    const double *pose_s = x + XO_POSE + 7 * s;
    const v3 p_w = qrot(ldq_pose(pose_s), p_i) + ld3(pose_s);
    auto rho_of = [&](const v3 &pcj, double px, double py) {
      const double inv_z = 1.0 / pcj.z;
      const double r0 = sq * (pcj.x * inv_z - px), r1 = sq * (pcj.y * inv_z - py);
      double rho[3];
      huber_rho(huber_a, r0 * r0 + r1 * r1, rho);
      return rho[0];
    };
    for (int t = 0; t < wv.kmax; ++t) {
      const unsigned char fl = flg[(size_t)t * n + lane];
      if (!(fl & 1)) continue;
      const double *ob = obs + (size_t)t * 11 * n + lane;
      const double dtj = td - ob[(size_t)10 * n];
      v3 p_j = p_i;
      if (t > 0) {
        const double *pose_j = x + XO_POSE + 7 * min(s + t, VILO_MAX_FRAMES - 1);
        p_j = qrot(qinv(ldq_pose(pose_j)), p_w - ld3(pose_j));
        const v3 pcj = qrot(qinv(qic), p_j - tic);
        cost += rho_of(pcj, ob[0] - ob[(size_t)6 * n] * dtj, ob[(size_t)1 * n] - ob[(size_t)7 * n] * dtj);
      }
      if (fl & 2) {
        const v3 pcj = qrot(qinv(qic2), p_j - tic2);
        cost += rho_of(pcj, ob[(size_t)3 * n] - ob[(size_t)8 * n] * dtj, ob[(size_t)4 * n] - ob[(size_t)9 * n] * dtj);
      }
    }
  }
  const double csum = wave_sum(ls.active ? cost : 0.0);
  double *cost_out = b.chunk_cost + (size_t)wave_id * VILO_MAX_FRAMES;
  if (lane < VILO_MAX_FRAMES) cost_out[lane] = (lane == 0) ? csum : 0.0;
}

// =================================================================================================
// IMU-leg factors
// =================================================================================================
#define IMU_LIN_STRIDE (31 * 39)
#define IMU_NTRI 496   // upper triangle of the 31 x 31 sqrt_info

// IMULegFactor linearisation (imu_leg_factor.cpp:173-386) in two kernels:
//   k_imu_raw        one THREAD per factor (the code is a scalar dependency chain, so lanes = factors gives 64-way SIMD): raw residual and
//                    31 x 38 local Jacobian, structural non-zeros only (the zeros are set once when the batch is created), stored
//                    ENTRY-MAJOR over the batch — b.imu_raw[(row * 39 + col) * NF + f] — so that the 64 lanes of a store are 64 neighbours
//   k_imu_linearize  one wave per factor, both products on the FP64 matrix cores (v_mfma_f64_16x16x4_f64):
//                      whitening Jw = U [J | r]   (U = sqrt_info, upper triangular 31 x 31; 32 x 48 x 32 padded, zero blocks skipped);
//                                                 operands straight from global memory, only the structurally non-zero entries of [J | r]
//                      Gram      G  = Jw^T Jw     (39 x 39: the factor's J^T J, J^T r and r^T r = its cost; upper tiles only)
// Operand layout of the instruction: A(16 x 4): lane l holds A[l % 16][l / 16]; B(4 x 16): lane l holds B[l / 16][l % 16];
// C/D(16 x 16): register r of lane l is C[(l / 16) + 4 r][l % 16]. The whitened block goes to HBM only for the marginalisation (mode 0).
#define IW_JS 48   // LDS row stride of Jw (conflict-free operand reads of the Gram pass)
// structural non-zeros of [J | r] (31 rows x 39 columns, bit c of entry r): union of the IMULegFactor and the embedded IMUFactor patterns
__device__ constexpr unsigned long long c_imu_nz[31] = {
    0x4000387fffULL, 0x4000387fffULL, 0x4000387fffULL, 0x4001c07038ULL, 0x4001c07038ULL, 0x4001c07038ULL, 0x400e007ff8ULL, 0x400e007ff8ULL,
    0x400e007ff8ULL, 0x407038fe3fULL, 0x407038fe3fULL, 0x407038fe3fULL, 0x438039703fULL, 0x438039703fULL, 0x438039703fULL, 0x40003a703fULL,
    0x40003a703fULL, 0x40003a703fULL, 0x40003c703fULL, 0x40003c703fULL, 0x40003c703fULL, 0x4070000e00ULL, 0x4070000e00ULL, 0x4070000e00ULL,
    0x4380007000ULL, 0x4380007000ULL, 0x4380007000ULL, 0x4400008000ULL, 0x4800010000ULL, 0x5000020000ULL, 0x6000040000ULL};

__global__ void __launch_bounds__(64) k_imu_raw(BatchDev b, double g_norm, int mode) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int NF = b.W * 10;
  if (f >= NF) return;
  const int win = f / 10, k = f % 10;
  const SolverState &st = b.st[win];
  if (lin_skip(st, mode) || b.imu_skip[f]) return;
  const PreintPrepared &pp = b.prep[f];
  const double *x = (mode ? b.xc : b.x) + (size_t)win * XSTRIDE;
  double *raw = b.imu_raw + f;
  double r[31];
  if (b.win[win].use_leg) {
    imu_leg_raw(pp.head, g_norm, x + XO_POSE + 7 * k, x + XO_SB + 9 * k, x + XO_LB + 4 * k, x + XO_POSE + 7 * (k + 1),
                x + XO_SB + 9 * (k + 1), x + XO_LB + 4 * (k + 1), r, true, raw, 39 * NF, NF);
#pragma unroll
    for (int i = 0; i < 31; ++i) raw[(size_t)(i * 39 + 38) * NF] = r[i];
  } else {
    // plain IMUFactor (estimator.cpp:1160-1171) inside the same 31 x 39 layout: rows 0..14, the frame-j blocks at column 19,
    // leg-bias columns and rows 15..30 stay zero (sqrt_info is embedded accordingly, k_embed_sqrt15)
    imu_raw(pp.head, g_norm, x + XO_POSE + 7 * k, x + XO_SB + 9 * k, x + XO_POSE + 7 * (k + 1), x + XO_SB + 9 * (k + 1), r, true, raw, 39 * NF, 19, NF);
#pragma unroll
    for (int i = 0; i < 15; ++i) raw[(size_t)(i * 39 + 38) * NF] = r[i];
  }
}

typedef double dbl2 __attribute__((ext_vector_type(2)));

// One wave per PAIR of consecutive factors (2 p, 2 p + 1; the same window): a lane's 16-byte load of an entry-major raw entry brings
// both factors' values, which halves the scattered 32-byte sectors the gather touches per factor.
// single: one wave per FACTOR (workgroup 2 p + h takes factor h of pair p; the pair's loads as before) — small batches, where the second
// factor of a pair only lengthens the wave's latency chain and half of the SIMDs have nothing to do
__global__ void __launch_bounds__(64) k_imu_linearize(BatchDev b, int mode, int single) {
  __shared__ double Jw[32 * IW_JS];
  // The entry-major raw Jacobians put the same entry of 8 consecutive factors in one 64-byte line, i.e. four pairs share every line they
  // gather from. Workgroup b runs on XCD b % 8 (its own L2): inside each group of 32 workgroups, the four that land on XCD x take the
  // pairs 4 x .. 4 x + 3, so a line is fetched from HBM by one L2 instead of four. (A permutation of [0, gridDim.x); ragged tail: identity.)
  int pair = blockIdx.x, hsel = -1;
  if (single) { hsel = pair & 1; pair >>= 1; }
  else if ((pair | 31) < (int)gridDim.x) pair = (pair & ~31) + 4 * (pair & 7) + ((pair >> 3) & 3);
  const int f0 = 2 * pair, win = f0 / 10;
  SolverState &st = b.st[win];
  if (lin_skip(st, mode)) return;
  const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
  const int NF = b.W * 10;
  // [J | r] tiles of both factors: up to 24 B values per lane and factor (structural non-zeros only), all loads in flight at once
  double bv[2][8][3];
  {
    const dbl2 *raw2 = (const dbl2 *)(b.imu_raw + f0);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int J = 0; J < 3; ++J) {
        const int q = 4 * kk + lk, col = 16 * J + lr;
        const bool nz = q < 31 && col < 39 && ((c_imu_nz[min(q, 30)] >> col) & 1ULL);
        dbl2 v = {0.0, 0.0};
        if (nz) v = raw2[((size_t)(q * 39 + col) * NF) >> 1];
        bv[0][kk][J] = v[0]; bv[1][kk][J] = v[1];
      }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (hsel >= 0 && h != hsel) continue;
    const int f = f0 + h;
    double *gout = b.imu_gram + (size_t)f * 780;
    if (b.imu_skip[f]) {   // no factor for this interval: it contributes nothing to the normal equations
      if (mode == 0) for (int e = lane; e < IMU_LIN_STRIDE; e += 64) b.imu_lin[(size_t)f * IMU_LIN_STRIDE + e] = 0.0;
      for (int e = lane; e < 780; e += 64) gout[e] = 0.0;
      if (lane == 0) b.imu_cost[f] = 0.0;
      continue;
    }
    const bool prof = (f % 10 == 0 && lane == 0);
    const long long c0 = pclk64();
    const double *U = b.prep[f].sqrt_info;
    // sqrt_info operands straight from global memory: 12 A values per lane
    double av[2][8];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int row = 16 * I + lr, q = 4 * kk + lk;
        av[I][kk] = (kk >= 4 * I && row < 31 && q < 31) ? U[row * 31 + q] : 0.0;   // U(16 .. 31, 0 .. 15) = 0: never loaded
      }
    const long long c1 = pclk64();
    double *out = b.imu_lin + (size_t)f * IMU_LIN_STRIDE;
    lds_fence();   // (the previous factor's Gram pass has read Jw)
#pragma unroll
    for (int I = 0; I < 2; ++I) {
#pragma unroll
      for (int J = 0; J < 3; ++J) {
        mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = (I == 0 ? 0 : 4); kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I][kk], bv[h][kk][J], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * I + lk + 4 * r, col = 16 * J + lr;
          Jw[row * IW_JS + col] = acc[r];   // padding rows / columns come out as exact zeros
          if (mode == 0 && row < 31 && col < 39) out[row * 39 + col] = acc[r];
        }
      }
    }
    lds_fence();
    const long long c2 = pclk64();
#pragma unroll
    for (int I = 0; I < 3; ++I) {
#pragma unroll
      for (int J = I; J < 3; ++J) {
        mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const double a_ = Jw[(4 * kk + lk) * IW_JS + 16 * I + lr];
          const double b_ = Jw[(4 * kk + lk) * IW_JS + 16 * J + lr];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = 16 * I + lk + 4 * r, bc = 16 * J + lr;
          if (a <= bc && bc < 39) gout[tri39(a, bc)] = acc[r];
          if (a == 38 && bc == 38) b.imu_cost[f] = acc[r];   // |sqrt_info r|^2
        }
      }
    }
    PCLK(if (prof) { const long long c3 = clock64(); st.phase_clk[33] = c1 - c0; st.phase_clk[34] = c2 - c1; st.phase_clk[35] = c3 - c2; });
  }
}

// k_imu_raw + k_imu_linearize of ONE factor in one wave, for small batches: the [J | r] block never leaves the wave. Lane 0 evaluates
// what depends on the states — the residual and seven 3 x 3 matrices (imu_blocks, the expressions of imu_leg_raw / imu_raw) — into a
// pool of 96 doubles; every lane then fetches the 24 entries of [J | r] its matrix-core operands hold from where the compile-time table
// says they come from (pool, the record's head, +-1, times 1 / -1 / T / -T: imu_gather_table), and the wave whitens and forms the Gram
// exactly as k_imu_linearize does (same operand order, same MFMA sequence). A single lane writing the ~300 entries one after the other
// took 17 k cycles alone on the chip and 35 k beside the visual waves; the block evaluation takes a fraction of that.
// The body takes its LDS from the caller, so the workgroups of a visual kernel's launch that are not packed waves can run it — an
// iteration of a small batch is a chain of kernel latencies, and this takes the two IMU launches out of the chain.
// lds: IMU_FUSED_LDS doubles = Jw 32 x 48 | PreintHead (126, padded to 128) | the two frames' states (40) | block pool (IB_N = 96)
__device__ const vilo::ImuGatherTable c_imu_gather[2] = {vilo::imu_gather_table(false), vilo::imu_gather_table(true)};
__device__ __forceinline__ void imu_fused_body(BatchDev &b, int f, double g_norm, int mode, double *lds) {
  static_assert(32 * IW_JS + 128 + 40 + IB_N == IMU_FUSED_LDS, "LDS of the fused IMU body");
  double *const Jw = lds, *const hd = lds + 32 * IW_JS, *const xl = hd + 128, *const pool = xl + 40;
  const int win = f / 10, k = f - 10 * win;
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  SolverState &st = b.st[win];
  // (the buffer the visual half of this pass writes its landmark gradients to: kept for the workgroups of the NEXT launch that finish the
  // frame-parallel form while the bookkeeping of their window flips st.cur — visual_reduce_body)
  if (k == 0 && lane == 0 && b.lin_cur) b.lin_cur[win] = st.cur;
  if (lin_skip(st, mode)) return;
  double *gout = b.imu_gram + (size_t)f * 780;
  if (b.imu_skip[f]) {   // no factor for this interval: it contributes nothing to the normal equations
    if (mode == 0) for (int e = lane; e < IMU_LIN_STRIDE; e += 64) b.imu_lin[(size_t)f * IMU_LIN_STRIDE + e] = 0.0;
    for (int e = lane; e < 780; e += 64) gout[e] = 0.0;
    if (lane == 0) b.imu_cost[f] = 0.0;
    return;
  }
  const PreintPrepared &pp = b.prep[f];
  const double *x = (mode ? b.xc : b.x) + (size_t)win * XSTRIDE;
  const bool leg = b.win[win].use_leg != 0;
  const long long fc0 = pclk64();
  // this lane's 24 entries of the gather table (operand layout of the whitening: row 4 kk + lk, column 16 J + lr)
  unsigned short gt[8][3];
  {
    const unsigned short *tab = c_imu_gather[leg ? 1 : 0].e;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int J = 0; J < 3; ++J) gt[kk][J] = tab[(4 * kk + lk) * 48 + 16 * J + lr];
  }
  {
    // stage the record's head and the two frames' states (coalesced)
    const double *hsrc = (const double *)&pp.head;
    const double h0 = hsrc[lane], h1 = (lane + 64 < 126) ? hsrc[lane + 64] : 0.0;
    double xv = 0.0;
    if (lane < 40) {
      // [pose_i 7 | sb_i 9 | lb_i 4 | pose_j 7 | sb_j 9 | lb_j 4]
      const int h = lane >= 20 ? 1 : 0, e = lane - 20 * h;
      xv = e < 7 ? x[XO_POSE + 7 * (k + h) + e] : (e < 16 ? x[XO_SB + 9 * (k + h) + (e - 7)] : x[XO_LB + 4 * (k + h) + (e - 16)]);
    }
    hd[lane] = h0; hd[lane + 64] = h1;
    if (lane < 40) xl[lane] = xv;
  }
  // sqrt_info operands straight from global memory (in flight under the block evaluation): 12 A values per lane
  const double *U = pp.sqrt_info;
  double av[2][8];
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int row = 16 * I + lr, q = 4 * kk + lk;
      av[I][kk] = (kk >= 4 * I && row < 31 && q < 31) ? U[row * 31 + q] : 0.0;   // U(16 .. 31, 0 .. 15) = 0: never loaded
    }
  lds_fence();
  const long long fc1 = pclk64();
  if (lane == 0) imu_blocks(*(const vilo::PreintHead *)hd, g_norm, leg, xl, xl + 7, xl + 16, xl + 20, xl + 27, xl + 36, pool);
  lds_fence();
  const long long fc2 = pclk64();
  // [J | r] operands of the whitening, each from its source
  const double T = hd[0];
  double bv[8][3];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk)
#pragma unroll
    for (int J = 0; J < 3; ++J) {
      const unsigned g_ = gt[kk][J], code = g_ >> 12;
      const double val = ((g_ & 0x100) ? hd : pool)[g_ & 0xff];
      const double cf = code == 1 ? 1.0 : (code == 2 ? -1.0 : (code == 3 ? T : -T));
      bv[kk][J] = code ? cf * val : 0.0;
    }
  lds_fence();   // (every operand is in registers before the whitened block overwrites the image)
  double *out = b.imu_lin + (size_t)f * IMU_LIN_STRIDE;
#pragma unroll
  for (int I = 0; I < 2; ++I) {
#pragma unroll
    for (int J = 0; J < 3; ++J) {
      mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = (I == 0 ? 0 : 4); kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I][kk], bv[kk][J], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + lk + 4 * r, col = 16 * J + lr;
        Jw[row * IW_JS + col] = acc[r];   // padding rows / columns come out as exact zeros
        if (mode == 0 && row < 31 && col < 39) out[row * 39 + col] = acc[r];
      }
    }
  }
  lds_fence();
#pragma unroll
  for (int I = 0; I < 3; ++I) {
#pragma unroll
    for (int J = I; J < 3; ++J) {
      mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const double a_ = Jw[(4 * kk + lk) * IW_JS + 16 * I + lr];
        const double b_ = Jw[(4 * kk + lk) * IW_JS + 16 * J + lr];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = 16 * I + lk + 4 * r, bc = 16 * J + lr;
        if (a <= bc && bc < 39) gout[tri39(a, bc)] = acc[r];
        if (a == 38 && bc == 38) b.imu_cost[f] = acc[r];   // |sqrt_info r|^2
      }
    }
  }
  PCLK(if (k == 0 && lane == 0) { const long long fc3 = clock64(); st.phase_clk[33] = fc1 - fc0; st.phase_clk[34] = fc2 - fc1; st.phase_clk[35] = fc3 - fc2; });
}

// Small batches with compact visual rows: the frame-parallel visual workgroups (one per (packed wave, frame)) and the IMU factors' (one
// per factor) in ONE launch. The IMU workgroups come first in the grid (they are the longer ones: started first, they end under the
// visual ones).
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_lin_small_c(BatchDev b, double sq, double huber_a, double gn, int mode, int n_imu) {
  constexpr int POOL = VisPool<true>::N > IMU_FUSED_LDS ? VisPool<true>::N : IMU_FUSED_LDS;
  __shared__ __attribute__((aligned(16))) double pool[POOL];
  const int bid = blockIdx.x;
  if (bid < n_imu) { imu_fused_body(b, bid, gn, mode, pool); return; }
  const int v = bid - n_imu;
  visual_linearize_body<true, true>(b, sq, huber_a, mode, pool, v % b.n_waves, v / b.n_waves);
}

// residual-only at the candidate (the last one of a solve). One wave per 64 factors: lane = factor for the raw residual (a scalar chain),
// then factor by factor lane = row of sqrt_info for |sqrt_info r|^2, the residual broadcast from LDS — sqrt_info is read where the
// preparation left it (row-major per record; a lane's row is 248 B: its lines stay in L1 across the 31 column steps), so no entry-major
// transpose of all 40 960 matrices per solve is needed for this one pass.
__global__ void __launch_bounds__(64) k_imu_cost(BatchDev b, double g_norm) {
  __shared__ double rs[32 * 64];   // [entry][factor of the wave]; row 31: the squared whitened entries summed per factor
  const int lane = threadIdx.x, f0 = blockIdx.x * 64, f = f0 + lane;
  const int NF = b.W * 10;
  bool live = false;
  {
    double r[31];
#pragma unroll
    for (int i = 0; i < 31; ++i) r[i] = 0.0;
    if (f < NF) {
      const int win = f / 10, k = f % 10;
      live = !b.st[win].done && !b.imu_skip[f];
      if (live) {
        const PreintPrepared &pp = b.prep[f];
        const double *x = b.xc + (size_t)win * XSTRIDE;
        if (b.win[win].use_leg) {
          imu_leg_raw(pp.head, g_norm, x + XO_POSE + 7 * k, x + XO_SB + 9 * k, x + XO_LB + 4 * k, x + XO_POSE + 7 * (k + 1),
                      x + XO_SB + 9 * (k + 1), x + XO_LB + 4 * (k + 1), r, false, nullptr, 0);
        } else {
          imu_raw(pp.head, g_norm, x + XO_POSE + 7 * k, x + XO_SB + 9 * k, x + XO_POSE + 7 * (k + 1), x + XO_SB + 9 * (k + 1), r, false, nullptr, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 31; ++i) rs[i * 64 + lane] = r[i];
    rs[31 * 64 + lane] = 0.0;
  }
  lds_barrier();
  const unsigned long long livem = __ballot(live);
  const int row = min(lane, 30);
  for (int fl = 0; fl < 64; ++fl) {
    if (!((livem >> fl) & 1ULL)) continue;
    const double *U = b.prep[f0 + fl].sqrt_info + row * 31;
    double u[31];
#pragma unroll
    for (int q = 0; q < 31; ++q) u[q] = (q >= row) ? U[q] : 0.0;
    double sacc = 0.0;
#pragma unroll
    for (int q = 0; q < 31; ++q) sacc += u[q] * rs[q * 64 + fl];
    const double c = wave_sum(lane < 31 ? sacc * sacc : 0.0);
    if (lane == 0) rs[31 * 64 + fl] = c;
  }
  lds_barrier();
  if (f < NF && !b.st[f / 10].done) b.imu_cost[f] = live ? rs[31 * 64 + lane] : 0.0;
}

__global__ void __launch_bounds__(256) k_embed_sqrt15(BatchDev b) {
  __shared__ double t[225];
  PreintPrepared &pp = b.prep[blockIdx.x];
  for (int e = threadIdx.x; e < 225; e += 256) t[e] = pp.sqrt_info[e];
  __syncthreads();
  for (int e = threadIdx.x; e < 31 * 31; e += 256) {
    const int i = e / 31, q = e - 31 * i;
    pp.sqrt_info[e] = (i < 15 && q < 15) ? t[i * 15 + q] : 0.0;
  }
}
int vilo_launch_embed_sqrt15(vilo_ctx *ctx, BatchDev &b) {
  hipLaunchKernelGGL(k_embed_sqrt15, dim3(b.W * 10), dim3(256), 0, ctx->stream, b);
  VILO_HIP(hipGetLastError());
  return VILO_OK;
}


// =================================================================================================
// k_accept: candidate cost, step quality, accept / reject (TrustRegionMinimizer::{IsStepSuccessful,
// HandleSuccessfulStep, HandleUnsuccessfulStep, HandleInvalidStep} + DoglegStrategy::Step{Accepted,Rejected,IsInvalid})
// =================================================================================================
__global__ void __launch_bounds__(128) k_accept(BatchDev b, AcceptParams ap) {
  __shared__ double red[128];
  __shared__ double dxs[VILO_MAX_PRIOR_DIM];
  __shared__ int accept_s;
  accept_body(b, ap, red, dxs, &accept_s);
}

__global__ void k_init_state(BatchDev b, double radius0, double mu0, int fail_bad) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= b.W) return;
  SolverState &s = b.st[w];
  memset(&s, 0, sizeof(SolverState));
  s.radius = radius0;
  s.mu = mu0;
  s.need_lin = 1;
  s.t_start = wall_clock64();
  if (fail_bad && b.win_bad && b.win_bad[w]) { s.done = 1; s.termination = 2; }
}

// =================================================================================================
// host-side launch sequence
// =================================================================================================
int vilo_launch_wave_solver(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s, int stage);   // kernels_wave.hip
int vilo_launch_assemble_small(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s, const AcceptParams *ap, int reduce_waves);   // kernels_asm_small.hip
bool vilo_assemble_small_takes(const BatchDev &b);                                                                              // kernels_asm_small.hip
int vilo_launch_split_stage(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s, int which);   // kernels_split.hip
int vilo_solver_form(const vilo_ctx *ctx, const BatchDev &b);                                                                    // kernels_wave.hip
int vilo_launch_assemble_full(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s, const AcceptParams &ap, int which);   // kernels_asm_full.hip

int vilo_solve_launch(vilo_ctx *ctx, BatchDev &b, const vilo_solve_opts *o) {
  const double sq = ctx->cfg.focal_length / 1.5, ha = ctx->cfg.huber_delta, gn = ctx->cfg.g_norm;
  hipStream_t s = ctx->stream;
  SolveParams sp;
  sp.min_lm_diagonal = o->min_lm_diagonal; sp.max_lm_diagonal = o->max_lm_diagonal;
  sp.min_radius = o->min_trust_region_radius; sp.gradient_tolerance = o->gradient_tolerance;
  sp.jacobi_scaling = o->jacobi_scaling; sp.fixed_iterations = o->fixed_iterations;
  AcceptParams ap;
  ap.min_relative_decrease = o->min_relative_decrease; ap.function_tolerance = o->function_tolerance;
  ap.parameter_tolerance = o->parameter_tolerance; ap.max_num_iterations = o->max_num_iterations;
  ap.fixed_iterations = o->fixed_iterations; ap.init_mode = 1; ap.max_solver_time_us = o->max_solver_time_us;
  const int W = b.W;
  int pidx = 0;
  ctx->pev_kind.clear();
  // profile: 0 off, 1 every kernel, 2 + k only kernel kind k (a pair of event records costs the stream ~7 us; ~80 pairs per solve)
  bool pev_open = false;
  auto P0 = [&](int kind) {
    pev_open = ctx->profile == 1 || ctx->profile == 2 + kind;
    if (!pev_open) return;
    while ((int)ctx->pev.size() < 2 * (pidx + 1)) { hipEvent_t e; (void)hipEventCreate(&e); ctx->pev.push_back(e); }
    ctx->pev_kind.push_back(kind);
    (void)hipEventRecord(ctx->pev[2 * pidx], s);
  };
  auto P1 = [&]() {
    if (!pev_open) return;
    (void)hipEventRecord(ctx->pev[2 * pidx + 1], s);
    ++pidx;
  };
  P0(6);
  hipLaunchKernelGGL(k_init_state, dim3((W + 127) / 128), dim3(128), 0, s, b, o->initial_trust_region_radius, ctx->initial_mu, 1);
  P1();
  // the first "candidate" is the initial point itself (IterationZero evaluates and linearises it)
  VILO_HIP(hipMemcpyAsync(b.xc, b.x, sizeof(double) * (size_t)W * XSTRIDE, hipMemcpyDeviceToDevice, s));
  if (b.n_lm > 0) VILO_HIP(hipMemcpyAsync(b.lamc, b.lam, sizeof(double) * (size_t)b.n_lm, hipMemcpyDeviceToDevice, s));
  for (int it = 0; it < o->max_num_iterations; ++it) {
    // cost + linearisation of the candidate -> accept / reject -> (accepted: normal equations) -> step -> next candidate
    if (b.rp_on) {
      P0(10);
      if (vilo_repropagate_launch(ctx, b, 1, 0) != VILO_OK) return VILO_ERR_HIP;
      P1();
      P0(11);
      if (vilo_repropagate_launch(ctx, b, 1, 1) != VILO_OK) return VILO_ERR_HIP;
      P1();
    }
    // Which assembly a batch gets (compact slots: td a constant block in every window — all of the reference's configurations):
    //   up to 256 windows (VILO_ASM_SMALL_MAX_WINDOWS): an iteration is a chain of kernel latencies, so the chain is kept short — the IMU
    //     factors are linearised by extra workgroups of the visual launch (imu_fused_body: k_imu_raw + k_imu_linearize of one factor in
    //     one wave, bitwise the same Gram), the bookkeeping, the assembly and the second half of the frame-parallel visual form share one
    //     launch of 768 threads per window (k_assemble_s): three launches per iteration (linearise, bookkeeping + assemble, solve);
    //   beyond: the assembly in two kernels by LDS footprint (kernels_asm_full.hip: the pose part at four workgroups per CU with the
    //     bookkeeping as its first phase — no k_accept launch, the prior's H never read —, the speed / leg-bias part at six).
    // The IMU workgroups inside the visual launch pay up to 2048 windows (768: + 5 %, 1024: + 2.6 %, 2048: + 0.9 %; at 4096 the two forms
    // take the same time and the full batch keeps its separate kernels): VILO_SMALL_FUSE_MAX_WINDOWS moves that threshold (0: never).
    static const int small_max = [] { const char *e = getenv("VILO_SMALL_FUSE_MAX_WINDOWS"); return e ? atoi(e) : 2048; }();
    const bool asm_small = vilo_assemble_small_takes(b);
    // (beyond the small form a batch with few packed waves — windows of a handful of landmarks — runs the frame-parallel visual form with
    // its own reduction kernel: only k_assemble_s has workgroups for that reduction)
    const bool fuse_imu = W <= small_max && visual_launch_takes_imu(b) && (asm_small || !b.lm_part);
    P0(0);
    launch_visual_linearize(b, sq, ha, s, 1, fuse_imu, gn);
    P1();
    // (the fused body as a kernel of its own for full batches — one wave per factor, no raw block through HBM — was measured slower than
    // the two kernels: 315 - 319 us against 45 + 245 at 4096 windows: there lanes = factors is the better form of the raw evaluation)
    if (!fuse_imu) {
      P0(7);
      hipLaunchKernelGGL(k_imu_raw, dim3((W * 10 + 63) / 64), dim3(64), 0, s, b, gn, 1);
      P1();
      P0(1);
      {
        static const int single_max = [] { const char *e = getenv("VILO_IMU_SINGLE_MAX_WINDOWS"); return e ? atoi(e) : 128; }();   // (measured: 128 windows + 1 %, 256 equal, 512 - 3 %)
        const int single = W <= single_max ? 1 : 0;   // (one wave per factor while the batch leaves SIMDs idle)
        hipLaunchKernelGGL(k_imu_linearize, dim3(W * (single ? 10 : 5)), dim3(64), 0, s, b, 1, single);
      }
      P1();
    }
    if (asm_small) {
      const bool reduce_later = fuse_imu && b.lm_part;   // (k_assemble_s's extra workgroups run visual_reduce_body)
      P0(8);
      if (vilo_launch_assemble_small(ctx, b, sp, s, &ap, reduce_later ? b.n_waves : 0) < 0) return VILO_ERR_HIP;
      P1();
    } else if (b.compact) {
      P0(8);
      if (vilo_launch_assemble_full(ctx, b, sp, s, ap, 0) != VILO_OK) return VILO_ERR_HIP;
      P1();
      P0(2);
      if (vilo_launch_assemble_full(ctx, b, sp, s, ap, 1) != VILO_OK) return VILO_ERR_HIP;
      P1();
    } else {
      // 23-column slots (a window estimates td): the bookkeeping as a kernel of its own, then k_assemble
      P0(5);
      hipLaunchKernelGGL(k_accept, dim3(W), dim3(128), 0, s, b, ap);
      P1();
      P0(8);
      if (vilo_launch_wave_solver(ctx, b, sp, s, 0) != VILO_OK) return VILO_ERR_HIP;
      P1();
    }
    ap.init_mode = 0;
    if (vilo_solver_form(ctx, b) == 3) {
      // three-stage form (kernels_split.hip): chain -> pose system -> back-substitutions + step, then the complete single-wave solver for
      // the windows a stage flagged (a factorisation failed: the retry loop lives there) — it returns at once for the rest
      P0(12);
      if (vilo_launch_split_stage(ctx, b, sp, s, 0) != VILO_OK) return VILO_ERR_HIP;
      P1();
      P0(13);
      if (vilo_launch_wave_solver(ctx, b, sp, s, 2) != VILO_OK) return VILO_ERR_HIP;
      P1();
      P0(14);
      if (vilo_launch_split_stage(ctx, b, sp, s, 1) != VILO_OK) return VILO_ERR_HIP;
      P1();
      P0(9);
      if (vilo_launch_wave_solver(ctx, b, sp, s, 4) != VILO_OK) return VILO_ERR_HIP;
      P1();
    } else {
      P0(9);
      if (vilo_launch_wave_solver(ctx, b, sp, s, 1) != VILO_OK) return VILO_ERR_HIP;
      P1();
    }
  }
  // the last candidate (or, without iterations, the initial point) only needs its cost
  P0(3);
  if (b.n_waves > 0) {
    // few packed waves: one workgroup per (packed wave, frame) fills the chip; many: one wave per packed wave reads every observation once
    if (b.lm_part) hipLaunchKernelGGL(k_visual_cost, dim3(b.n_waves, VILO_MAX_FRAMES), dim3(64), 0, s, b, sq, ha);
    else hipLaunchKernelGGL(k_visual_cost_walk, dim3(b.n_waves), dim3(64), 0, s, b, sq, ha);
  }
  P1();
  if (b.rp_on) {
    P0(10);
    if (vilo_repropagate_launch(ctx, b, 1, 0) != VILO_OK) return VILO_ERR_HIP;
    P1();
    P0(11);
    if (vilo_repropagate_launch(ctx, b, 1, 1) != VILO_OK) return VILO_ERR_HIP;
    P1();
  }
  P0(4);
  hipLaunchKernelGGL(k_imu_cost, dim3((W * 10 + 63) / 64), dim3(64), 0, s, b, gn);
  P1();
  P0(5);
  hipLaunchKernelGGL(k_accept, dim3(W), dim3(128), 0, s, b, ap);
  P1();
  VILO_HIP(hipGetLastError());
  return VILO_OK;
}

// preMarginalize (marginalization_factor.cpp:119-138): evaluate the factors once at the current state.
int vilo_marg_linearize(vilo_ctx *ctx, BatchDev &b) {
  const double sq = ctx->cfg.focal_length / 1.5, ha = ctx->cfg.huber_delta, gn = ctx->cfg.g_norm;
  hipLaunchKernelGGL(k_init_state, dim3((b.W + 127) / 128), dim3(128), 0, ctx->stream, b, 1e4, 1e-8, 0);
  if (b.rp_on && (vilo_repropagate_launch(ctx, b, 0, 0) != VILO_OK || vilo_repropagate_launch(ctx, b, 0, 1) != VILO_OK)) return VILO_ERR_HIP;
  launch_visual_linearize(b, sq, ha, ctx->stream, 0);
  hipLaunchKernelGGL(k_imu_raw, dim3((b.W * 10 + 63) / 64), dim3(64), 0, ctx->stream, b, gn, 0);
  hipLaunchKernelGGL(k_imu_linearize, dim3(b.W * 5), dim3(64), 0, ctx->stream, b, 0, 0);
  VILO_HIP(hipGetLastError());
  return VILO_OK;
}
