// Fused sliding-window solver for gfx950: what ceres::Solve does for Estimator::optimization()
// (estimator.cpp:1054-1245; DENSE_SCHUR + traditional DOGLEG, Ceres 1.14 semantics) as a batched kernel
// pipeline over independent windows. Jacobians are never materialised in HBM:
//
//   k_visual_linearize  one wave per (window, start-frame) landmark chunk, lane = landmark. Evaluates the
//                       Projection*Factor residual blocks + Huber corrector in registers, reduces the landmark's
//                       1x1 Hessian / gradient / camera coupling row in registers, and forms the camera-side Gram
//                       blocks of every (start, t) pair through an LDS-staged wave reduction.
//   k_imu_raw           one thread per IMULegFactor: raw residual + 31x38 Jacobian (structural non-zeros only).
//   k_imu_whiten        one wave per IMULegFactor: whitening by the hoisted sqrt_info and the factor's 39x39 Gram, both
//                       on the FP64 matrix cores (v_mfma_f64_16x16x4_f64).
//   k_build_solve       one workgroup per window, everything LDS-resident: assembles the block-arrow camera system
//                       (dense 80x80 pose/extrinsic/td part + block-tridiagonal speed-bias/leg-bias part), Jacobi
//                       scaling, dogleg quantities, landmark Schur complement, block elimination, dense Cholesky,
//                       back-substitution, dogleg step, candidate state.
//   k_visual_cost / k_imu_cost / k_accept   trial-point cost and the trust-region accept/reject logic.
#include "solve_common.hpp"

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
#define XLANE 54  // LDS stride per lane: 2 rows x 26 cols + 2 pad; even so that every row starts 16-byte aligned (ds_read_b128)

// Per-lane view of a packed wave (WaveMeta): which chunk (start frame) a lane belongs to.
struct LaneSeg {
  int seg, s, gi, li;   // segment (-1: padding lane), start frame, global / window-local landmark index
  bool active;
};
__device__ __forceinline__ LaneSeg lane_segment(const WaveMeta &wv, const ChunkMeta *chunks, int lane, int cs[4], int cn[4], int ckm[4], int cgo[4]) {
  LaneSeg ls;
  ls.seg = -1; ls.s = 0; ls.gi = 0; ls.li = 0; ls.active = false;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    cs[g] = 0; cn[g] = 0; ckm[g] = 0; cgo[g] = 0;
    if (g < wv.nseg) {
      const ChunkMeta cm = chunks[wv.seg_chunk[g]];
      cs[g] = cm.s; cn[g] = cm.n; ckm[g] = cm.kmax; cgo[g] = cm.gram_off;
      const int i = lane - wv.seg_lane0[g];
      if (i >= 0 && i < cm.n) { ls.seg = g; ls.s = cm.s; ls.gi = cm.lm_off + i; ls.li = cm.lm_local + i; ls.active = true; }
    }
  }
  return ls;
}

// TPAR = false: one wave per packed wave walks all its frames (the throughput form: landmark-side sums stay in registers).
// TPAR = true (small batches, b.lm_part != null): one wave per (packed wave, frame offset) so that a handful of windows still
// fills the chip; every landmark-side term is written per (frame, camera) and k_visual_reduce adds the terms in the order the
// walking form adds them — the two forms give bitwise the same linearisation.
#define LM_NTERM 21   // E, g, w_pose_s (6), w_ex0 (6), w_ex1 (6), w_td
template <bool TPAR>
__device__ __forceinline__ void visual_linearize_body(BatchDev &b, double sq, double huber_a) {
  __shared__ __attribute__((aligned(16))) double X[(64 + 4) * XLANE + 16];   // 4 zero pad lanes = 8 pad rows
  __shared__ double xs[XSTRIDE];   // the window's state: poses are indexed per lane (lanes of a wave have different start frames)
  const int wave_id = b.wave_order[blockIdx.x];
  const WaveMeta wv = b.wave[wave_id];
  SolverState &st = b.st[wv.win];
  if (st.done || !st.need_lin) return;
  const WinMeta wm = b.win[wv.win];
  const int lane = threadIdx.x;
  int cs[4], cn[4], ckm[4], cgo[4];
  const LaneSeg ls = lane_segment(wv, b.chunk, lane, cs, cn, ckm, cgo);
  const bool active = ls.active;
  const int n = wv.n_lanes, L = wm.L, s = ls.s;
  const bool prof = !TPAR && (wave_id == wm.wave_off) && lane == 0;
  long long c_proj = 0, c_gram = 0, c_t0 = clock64(), c_a = 0;
  const double *xg = b.x + (size_t)wv.win * XSTRIDE;
  double *wbase = b.lm_w + 80 * (size_t)wm.lm_off;
  const int li = ls.li;
  for (int e = lane; e < XSTRIDE; e += 64) xs[e] = xg[e];
  const double *x = xs;

  // Gram of a (start frame, t) slot on the FP64 matrix cores: X^T X with X = the 2 n corrected Jacobian rows (26 columns,
  // padded to 32) as three 16 x 16 tiles (0,0), (0,1), (1,1). One k-step = 4 rows = 2 landmarks; lane (lr, lk) supplies
  // X[row 4 kk + lk][lr] (tile column 0) and X[..][16 + lr] (tile column 1), which serve as A and B operands alike.
  // Every segment of the wave (its own start frame) accumulates into its own three tiles.
  const int lr = lane & 15, lk = lane >> 4;
  const int xoff = (lk >> 1) * XLANE + (lk & 1) * 26 + lr;
  const bool c1on = lr < 10;   // columns 26 .. 31 of the second tile column do not exist
  if (!TPAR && active)   // TPAR: the host clears w before the launch (the frames of a landmark run in different workgroups)
    for (int a = 0; a < 80; ++a) wbase[(size_t)a * L + li] = 0.0;
  for (int e = lane; e < 4 * XLANE + 16; e += 64) X[64 * XLANE + e] = 0.0;

  const double *obs = b.obs + wv.obs_off;
  const unsigned char *flg = b.flags + wv.flag_off;
  double o12[12];
  double lam = 1.0;
  for (int c = 0; c < 12; ++c) o12[c] = 0.0;
  if (active) {
    lam = b.lam[ls.gi];
    o12[0] = obs[(size_t)0 * n + lane]; o12[1] = obs[(size_t)1 * n + lane]; o12[2] = obs[(size_t)2 * n + lane];
    o12[6] = obs[(size_t)6 * n + lane]; o12[7] = obs[(size_t)7 * n + lane]; o12[10] = obs[(size_t)10 * n + lane];
  }
  lds_barrier();
  const double *pose_s = x + XO_POSE + 7 * s, *ex0 = x + XO_EX, *ex1 = x + XO_EX + 7;
  const double td = x[XO_TD];

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
  const int t_begin = TPAR ? (int)blockIdx.y : 0, t_end = TPAR ? min((int)blockIdx.y + 1, wv.kmax) : wv.kmax;
  if (TPAR && t_begin > 0 && active && t_begin < wv.kmax) {   // this workgroup's frame instead of frame 0
    fl_next = flg[(size_t)t_begin * n + lane];
    const double *obn = obs + (size_t)t_begin * 11 * n;
#pragma unroll
    for (int c = 0; c < 11; ++c) on[c] = obn[(size_t)c * n + lane];
  }
  for (int t = t_begin; t < t_end; ++t) {
    const int j = min(s + t, VILO_MAX_FRAMES - 1);
    const unsigned char fl = fl_next;
    const double *pose_j = x + XO_POSE + 7 * j;
    double ob[11];
#pragma unroll
    for (int c = 0; c < 11; ++c) ob[c] = on[c];
    if (!TPAR && active && t + 1 < wv.kmax) {
      fl_next = flg[(size_t)(t + 1) * n + lane];
      const double *obn = obs + (size_t)(t + 1) * 11 * n;
#pragma unroll
      for (int c = 0; c < 11; ++c) on[c] = obn[(size_t)c * n + lane];
    }
    mfma_d4 G00[4], G01[4], G11[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { G00[g] = mfma_d4{0.0, 0.0, 0.0, 0.0}; G01[g] = G00[g]; G11[g] = G00[g]; }
    double wj[6];
    for (int c = 0; c < 6; ++c) wj[c] = 0.0;

    for (int cam = (t == 0 ? 1 : 0); cam < 2; ++cam) {
      // cam 0: left observation (TwoFrameOneCam); cam 1: right observation (TwoFrameTwoCam, or OneFrameTwoCam at t == 0)
      const bool produce = active && (fl & 1) && (cam == 0 || (fl & 2));
      double *xr0 = &X[lane * XLANE], *xr1 = xr0 + 26;
      c_a = clock64();
      if (produce) {
        double r[2], Ji[12], Jj[12], Je0[12], Je1[12], Jl[2], Jt[2];
        for (int c = 0; c < 12; ++c) Ji[c] = Jj[c] = Je0[c] = Je1[c] = 0.0;
        if (cam == 0) { o12[3] = ob[0]; o12[4] = ob[1]; o12[5] = ob[2]; o12[8] = ob[6]; o12[9] = ob[7]; }
        else { o12[3] = ob[3]; o12[4] = ob[4]; o12[5] = ob[5]; o12[8] = ob[8]; o12[9] = ob[9]; }
        o12[11] = ob[10];
        if (cam == 0) proj_factor<0>(o12, pose_s, pose_j, ex0, ex1, lam, td, sq, r, true, Ji, Jj, Je0, Je1, Jl, Jt);
        else if (t > 0) proj_factor<1>(o12, pose_s, pose_j, ex0, ex1, lam, td, sq, r, true, Ji, Jj, Je0, Je1, Jl, Jt);
        else proj_factor<2>(o12, pose_s, pose_j, ex0, ex1, lam, td, sq, r, true, Ji, Jj, Je0, Je1, Jl, Jt);
        const Corrector cr = make_corrector(huber_a, r[0] * r[0] + r[1] * r[1]);
        cost += cr.rho0;
        for (int c = 0; c < 6; ++c) {
          correct_col(cr, r[0], r[1], Ji[c], Ji[6 + c]);
          correct_col(cr, r[0], r[1], Jj[c], Jj[6 + c]);
          correct_col(cr, r[0], r[1], Je0[c], Je0[6 + c]);
          correct_col(cr, r[0], r[1], Je1[c], Je1[6 + c]);
        }
        correct_col(cr, r[0], r[1], Jl[0], Jl[1]);
        correct_col(cr, r[0], r[1], Jt[0], Jt[1]);
        r[0] *= cr.residual_scaling;
        r[1] *= cr.residual_scaling;
        // landmark-side reductions (the e-block of Ceres' Schur eliminator): the same 21 terms in both forms
        double term[LM_NTERM];
        term[0] = Jl[0] * Jl[0] + Jl[1] * Jl[1];
        term[1] = Jl[0] * r[0] + Jl[1] * r[1];
        for (int c = 0; c < 6; ++c) {
          term[2 + c] = Ji[c] * Jl[0] + Ji[6 + c] * Jl[1];
          term[8 + c] = Je0[c] * Jl[0] + Je0[6 + c] * Jl[1];
          term[14 + c] = Je1[c] * Jl[0] + Je1[6 + c] * Jl[1];
          wj[c] += Jj[c] * Jl[0] + Jj[6 + c] * Jl[1];
        }
        term[20] = Jt[0] * Jl[0] + Jt[1] * Jl[1];
        if (TPAR) {
          double *pt = b.lm_part + ((size_t)(t * 2 + cam) * LM_NTERM) * b.n_lm + ls.gi;
#pragma unroll
          for (int v = 0; v < LM_NTERM; ++v) pt[(size_t)v * b.n_lm] = term[v];
        } else {
          E += term[0];
          gl += term[1];
          for (int c = 0; c < 6; ++c) { wc_s[c] += term[2 + c]; wc_e0[c] += term[8 + c]; wc_e1[c] += term[14 + c]; }
          wc_td += term[20];
        }
        for (int c = 0; c < 6; ++c) {
          xr0[c] = Ji[c]; xr1[c] = Ji[6 + c];
          xr0[6 + c] = Jj[c]; xr1[6 + c] = Jj[6 + c];
          xr0[12 + c] = Je0[c]; xr1[12 + c] = Je0[6 + c];
          xr0[18 + c] = Je1[c]; xr1[18 + c] = Je1[6 + c];
        }
        xr0[24] = Jt[0]; xr1[24] = Jt[1];
        xr0[25] = r[0]; xr1[25] = r[1];
      } else {
        for (int c = 0; c < 26; ++c) { xr0[c] = 0.0; xr1[c] = 0.0; }
      }
      lds_barrier();
      { const long long c_b = clock64(); c_proj += c_b - c_a; c_a = c_b; }
      // rows of padding / unobserved lanes are zero, and every segment spans a multiple of 8 lanes = 4 k-steps:
      // 8 LDS reads in flight, then 12 MFMAs per trip
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g >= wv.nseg || t >= ckm[g]) continue;
        const int k0 = wv.seg_lane0[g] >> 1, k1 = k0 + (((cn[g] + 7) & ~7) >> 1);
        for (int kk0 = k0; kk0 < k1; kk0 += 4) {
          double a0[4], a1[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double *xr = &X[2 * (kk0 + u) * XLANE + xoff];
            a0[u] = xr[0];
            a1[u] = xr[16];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double b1 = c1on ? a1[u] : 0.0;
            G00[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], a0[u], G00[g], 0, 0, 0);
            G01[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1, G01[g], 0, 0, 0);
            G11[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b1, G11[g], 0, 0, 0);
          }
        }
      }
      lds_barrier();
      c_gram += clock64() - c_a;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g >= wv.nseg || t >= ckm[g]) continue;
      double *gs = b.gram + (size_t)(cgo[g] + t) * VILO_GRAM;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lk + 4 * r;
        if (row <= lr) gs[tri26(row, lr)] = G00[g][r];
        if (c1on) gs[tri26(row, 16 + lr)] = G01[g][r];
        if (c1on && row < 10 && row <= lr) gs[tri26(16 + row, 16 + lr)] = G11[g][r];
      }
    }
    if (active && t > 0 && (fl & 1))
      for (int c = 0; c < 6; ++c) wbase[(size_t)(6 * j + c) * L + li] = wj[c];
  }
  if (!TPAR && active) {
    b.lm_E[ls.gi] = E;
    b.lm_g[ls.gi] = gl;
    for (int c = 0; c < 6; ++c) {
      wbase[(size_t)(6 * s + c) * L + li] = wc_s[c];
      wbase[(size_t)(CD_EX0 + c) * L + li] = wc_e0[c];
      wbase[(size_t)(CD_EX1 + c) * L + li] = wc_e1[c];
    }
    wbase[(size_t)CD_TD * L + li] = wc_td;
  }
  (void)cost;   // the cost at the linearisation point is k_visual_cost's job
  if (prof) { st.phase_clk[16] = clock64() - c_t0; st.phase_clk[17] = c_proj; st.phase_clk[18] = c_gram; st.phase_clk[19] = wv.n_lanes; st.phase_clk[20] = wv.kmax; }
}

__global__ void __launch_bounds__(64) k_visual_linearize(BatchDev b, double sq, double huber_a) { visual_linearize_body<false>(b, sq, huber_a); }
__global__ void __launch_bounds__(64) k_visual_linearize_tpar(BatchDev b, double sq, double huber_a) { visual_linearize_body<true>(b, sq, huber_a); }

// Second half of the TPAR form: per landmark, the (frame, camera) terms in the order the walking form adds them (frames ascending,
// left camera before right; an unobserved factor contributed +0.0).
__global__ void __launch_bounds__(64) k_visual_reduce(BatchDev b) {
  const WaveMeta wv = b.wave[blockIdx.x];
  const SolverState &st = b.st[wv.win];
  if (st.done || !st.need_lin) return;
  const WinMeta wm = b.win[wv.win];
  const int lane = threadIdx.x;
  int cs[4], cn[4], ckm[4], cgo[4];
  const LaneSeg ls = lane_segment(wv, b.chunk, lane, cs, cn, ckm, cgo);
  if (!ls.active) return;
  double acc[LM_NTERM];
#pragma unroll
  for (int v = 0; v < LM_NTERM; ++v) acc[v] = 0.0;
  for (int t = 0; t < wv.kmax; ++t)
    for (int cam = (t == 0 ? 1 : 0); cam < 2; ++cam) {
      const double *pt = b.lm_part + ((size_t)(t * 2 + cam) * LM_NTERM) * b.n_lm + ls.gi;
#pragma unroll
      for (int v = 0; v < LM_NTERM; ++v) acc[v] += pt[(size_t)v * b.n_lm];
    }
  double *wbase = b.lm_w + 80 * (size_t)wm.lm_off;
  const int L = wm.L, li = ls.li, s = ls.s;
  b.lm_E[ls.gi] = acc[0];
  b.lm_g[ls.gi] = acc[1];
  for (int c = 0; c < 6; ++c) {
    wbase[(size_t)(6 * s + c) * L + li] = acc[2 + c];
    wbase[(size_t)(CD_EX0 + c) * L + li] = acc[8 + c];
    wbase[(size_t)(CD_EX1 + c) * L + li] = acc[14 + c];
  }
  wbase[(size_t)CD_TD * L + li] = acc[20];
}

// First part of the TPAR form: what the walking form does before its frame loop — the coupling rows of the landmarks of every window
// that is about to be re-linearised start from zero (windows that keep their linearisation after a rejected step are not touched).
__global__ void __launch_bounds__(64) k_visual_clear(BatchDev b) {
  const WaveMeta wv = b.wave[blockIdx.x];
  const SolverState &st = b.st[wv.win];
  if (st.done || !st.need_lin) return;
  const WinMeta wm = b.win[wv.win];
  int cs[4], cn[4], ckm[4], cgo[4];
  const LaneSeg ls = lane_segment(wv, b.chunk, threadIdx.x, cs, cn, ckm, cgo);
  if (!ls.active) return;
  double *wbase = b.lm_w + 80 * (size_t)wm.lm_off;
  for (int a = 0; a < 80; ++a) wbase[(size_t)a * wm.L + ls.li] = 0.0;
}

// both forms behind one call (kernel kind 0 of the profiling table)
static void launch_visual_linearize(BatchDev &b, double sq, double ha, hipStream_t s) {
  if (b.n_waves <= 0) return;
  if (b.lm_part) {
    hipLaunchKernelGGL(k_visual_clear, dim3(b.n_waves), dim3(64), 0, s, b);
    (void)hipMemsetAsync(b.lm_part, 0, sizeof(double) * (size_t)VILO_MAX_FRAMES * 2 * LM_NTERM * b.n_lm, s);
    hipLaunchKernelGGL(k_visual_linearize_tpar, dim3(b.n_waves, VILO_MAX_FRAMES), dim3(64), 0, s, b, sq, ha);
    hipLaunchKernelGGL(k_visual_reduce, dim3(b.n_waves), dim3(64), 0, s, b);
  } else {
    hipLaunchKernelGGL(k_visual_linearize, dim3(b.n_waves), dim3(64), 0, s, b, sq, ha);
  }
}

// Residual-only evaluation at the candidate point (TrustRegionMinimizer::ComputeCandidatePointAndEvaluateCost).
// Also forms the candidate inverse depth: lambda_c = lambda - a * g_l / dhat_l^2 - b * y_l.
// One workgroup per (packed wave, frame offset t): 33 k short waves instead of 3 k waves walking up to 11 frames each — the pass
// is latency-bound, the frames of a landmark are independent here, and k_accept adds the per-(wave, t) partial sums.
__global__ void __launch_bounds__(64) k_visual_cost(BatchDev b, double sq, double huber_a, int init_mode) {
  const int wave_id = xcd_balanced(blockIdx.x, gridDim.x);
  const WaveMeta wv = b.wave[wave_id];
  const SolverState &st = b.st[wv.win];
  if (st.done || (!init_mode && !st.step_valid)) return;
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
    double lam = b.lam[gi];
    if (!init_mode) lam += -st.coef_a * b.lm_g[gi] / b.lm_dh2[gi] - st.coef_b * b.lm_y[gi];
    if (t == 0) b.lamc[gi] = lam;
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

// =================================================================================================
// IMU-leg factors
// =================================================================================================
#define IMU_LIN_STRIDE (31 * 39)
#define IMU_NTRI 496   // upper triangle of the 31 x 31 sqrt_info

// Stage 1 of the IMULegFactor linearisation: one THREAD per factor evaluates the raw residual and the 31 x 38 local
// Jacobian (imu_leg_factor.cpp:173-386 before whitening). The code is a long scalar dependency chain, so lanes = factors
// gives 64-way SIMD instead of one busy lane per wave. Only the structural non-zeros of [J | r] (31 x 39, row-major in
// b.imu_raw) are written; the zeros are set once when the batch is created.
__global__ void __launch_bounds__(64) k_imu_raw(BatchDev b, double g_norm) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= b.W * 10) return;
  const int win = f / 10, k = f % 10;
  const SolverState &st = b.st[win];
  if (st.done || !st.need_lin || b.imu_skip[f]) return;
  const PreintPrepared &pp = b.prep[f];
  const double *x = b.x + (size_t)win * XSTRIDE;
  double *raw = b.imu_raw + (size_t)f * IMU_LIN_STRIDE;
  double r[31];
  if (b.win[win].use_leg) {
    imu_leg_raw(pp.head, g_norm, x + XO_POSE + 7 * k, x + XO_SB + 9 * k, x + XO_LB + 4 * k, x + XO_POSE + 7 * (k + 1),
                x + XO_SB + 9 * (k + 1), x + XO_LB + 4 * (k + 1), r, true, raw, 39);
#pragma unroll
    for (int i = 0; i < 31; ++i) raw[i * 39 + 38] = r[i];
  } else {
    // plain IMUFactor (estimator.cpp:1160-1171) inside the same 31 x 39 layout: rows 0..14, the frame-j blocks at column 19,
    // leg-bias columns and rows 15..30 stay zero (sqrt_info is embedded accordingly, k_embed_sqrt15)
    imu_raw(pp.head, g_norm, x + XO_POSE + 7 * k, x + XO_SB + 9 * k, x + XO_POSE + 7 * (k + 1), x + XO_SB + 9 * (k + 1), r, true, raw, 39, 19);
#pragma unroll
    for (int i = 0; i < 15; ++i) raw[i * 39 + 38] = r[i];
  }
}

// Stage 2: one wave per factor, both products on the FP64 matrix cores (v_mfma_f64_16x16x4_f64):
//   whitening  Jw = U [J | r]      (U = sqrt_info, upper triangular 31 x 31; 32 x 48 x 32 padded, zero blocks skipped)
//   Gram       G  = Jw^T Jw        (39 x 39, the factor's J^T J, J^T r and r^T r; upper tiles only)
// Operand layout of the instruction: A(16 x 4): lane l holds A[l % 16][l / 16]; B(4 x 16): lane l holds B[l / 16][l % 16];
// C/D(16 x 16): register r of lane l is C[(l / 16) + 4 r][l % 16].
#define IW_JS 48   // LDS row stride of Jw (conflict-free operand reads of the Gram pass)

__global__ void __launch_bounds__(64) k_imu_whiten(BatchDev b) {
  __shared__ double Jw[32 * IW_JS];
  const int f = blockIdx.x, win = f / 10;
  SolverState &st = b.st[win];
  if (st.done || !st.need_lin) return;
  const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
  if (b.imu_skip[f]) {   // no factor for this interval: it contributes nothing to the normal equations
    for (int e = lane; e < IMU_LIN_STRIDE; e += 64) b.imu_lin[(size_t)f * IMU_LIN_STRIDE + e] = 0.0;
    for (int e = lane; e < 780; e += 64) b.imu_gram[(size_t)f * 780 + e] = 0.0;
    return;
  }
  const bool prof = (f % 10 == 0 && lane == 0);
  const long long c0 = clock64();
  const double *U = b.prep[f].sqrt_info;
  const double *raw = b.imu_raw + (size_t)f * IMU_LIN_STRIDE;
  // MFMA operands straight from global memory: 12 A values (U tiles) and 24 B values ([J | r] tiles) per lane, all
  // loads independent and issued before the first MFMA (one memory round trip, no LDS staging of the inputs)
  double av[2][8], bv[8][3];
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int row = 16 * I + lr, q = 4 * kk + lk;
      av[I][kk] = (kk >= 4 * I && row < 31 && q < 31) ? U[row * 31 + q] : 0.0;   // U(16 .. 31, 0 .. 15) = 0: never loaded
    }
#pragma unroll
  for (int kk = 0; kk < 8; ++kk)
#pragma unroll
    for (int J = 0; J < 3; ++J) {
      const int q = 4 * kk + lk, col = 16 * J + lr;
      bv[kk][J] = (q < 31 && col < 39) ? raw[q * 39 + col] : 0.0;
    }
  const long long c1 = clock64();
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
        if (row < 31 && col < 39) out[row * 39 + col] = acc[r];
      }
    }
  }
  lds_barrier();
  const long long c2 = clock64();
  double *gout = b.imu_gram + (size_t)f * 780;
#pragma unroll
  for (int I = 0; I < 3; ++I) {
#pragma unroll
    for (int J = I; J < 3; ++J) {
      mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const double av = Jw[(4 * kk + lk) * IW_JS + 16 * I + lr];
        const double bv = Jw[(4 * kk + lk) * IW_JS + 16 * J + lr];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = 16 * I + lk + 4 * r, bc = 16 * J + lr;
        if (a <= bc && bc < 39) gout[tri39(a, bc)] = acc[r];
      }
    }
  }
  if (prof) { const long long c3 = clock64(); st.phase_clk[24] = c1 - c0; st.phase_clk[25] = c2 - c1; st.phase_clk[26] = c3 - c2; }
}

// residual-only at the candidate: one thread per factor; sqrt_info is read from its entry-major transpose
// (b.sqrtT[e][factor], coalesced across the lanes of a wave)
__global__ void __launch_bounds__(64) k_imu_cost(BatchDev b, double g_norm, int init_mode) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int NF = b.W * 10;
  if (f >= NF) return;
  const int win = f / 10, k = f % 10;
  const SolverState &st = b.st[win];
  if (st.done || (!init_mode && !st.step_valid)) return;
  if (b.imu_skip[f]) { b.imu_cost[f] = 0.0; return; }
  const PreintPrepared &pp = b.prep[f];
  const double *x = b.xc + (size_t)win * XSTRIDE;
  double r[31];
  if (b.win[win].use_leg) {
    imu_leg_raw(pp.head, g_norm, x + XO_POSE + 7 * k, x + XO_SB + 9 * k, x + XO_LB + 4 * k, x + XO_POSE + 7 * (k + 1),
                x + XO_SB + 9 * (k + 1), x + XO_LB + 4 * (k + 1), r, false, nullptr, 0);
  } else {
#pragma unroll
    for (int i = 15; i < 31; ++i) r[i] = 0.0;
    imu_raw(pp.head, g_norm, x + XO_POSE + 7 * k, x + XO_SB + 9 * k, x + XO_POSE + 7 * (k + 1), x + XO_SB + 9 * (k + 1), r, false, nullptr, 0);
  }
  const double *ut = b.sqrtT + f;
  double c = 0.0;
  int e = 0;
#pragma unroll
  for (int i = 0; i < 31; ++i) {
    double sacc = 0.0;
#pragma unroll
    for (int q = i; q < 31; ++q) { sacc += ut[(size_t)e * NF] * r[q]; ++e; }
    c += sacc * sacc;
  }
  b.imu_cost[f] = c;
}

// use_leg == 0: the 15 x 15 sqrt_info of an IMUFactor (leading 225 doubles, row stride 15) re-laid out as the leading block of a
// 31 x 31 matrix (row stride 31, zeros elsewhere) so that the IMU-leg kernels can be used unchanged
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

// entry-major transpose of the upper triangles of sqrt_info (once per batch)
__global__ void k_sqrt_transpose(BatchDev b) {
  const int f = blockIdx.x, NF = b.W * 10;
  const PreintPrepared &pp = b.prep[f];
  for (int e = threadIdx.x; e < 31 * 31; e += blockDim.x) {
    const int i = e / 31, q = e - 31 * i;
    if (q >= i) b.sqrtT[(size_t)(i * 31 - (i * (i - 1)) / 2 + (q - i)) * NF + f] = pp.sqrt_info[e];
  }
}
int vilo_launch_sqrt_transpose(vilo_ctx *ctx, BatchDev &b) {
  hipLaunchKernelGGL(k_sqrt_transpose, dim3(b.W * 10), dim3(64), 0, ctx->stream, b);
  VILO_HIP(hipGetLastError());
  return VILO_OK;
}

// =================================================================================================
// k_build_solve
// =================================================================================================
// camera dim of column c (0..37) of IMULegFactor(k, k+1)'s local Jacobian
__device__ __forceinline__ int imu_col_cd(int k, int c) {
  if (c < 6) return 6 * k + c;
  if (c < 19) return CD_B0 + 13 * k + (c - 6);
  if (c < 25) return 6 * (k + 1) + (c - 19);
  return CD_B0 + 13 * (k + 1) + (c - 25);
}

#define SOLVE_THREADS 256
#define CLD 81   // leading dimension of the 80x80 pose system in LDS (odd: conflict-free row and column walks)
#define LDS_C 0
#define LDS_AD (LDS_C + 80 * CLD)
#define LDS_AO (LDS_AD + 11 * 169)
#define LDS_G (LDS_AO + 10 * 169 + 1)
#define LDS_DH2 (LDS_G + CD_N)
#define LDS_Y (LDS_DH2 + CD_N)
#define LDS_TMP (LDS_Y + CD_N)
#define LDS_ACT (LDS_TMP + CD_N)
#define LDS_LK (LDS_ACT + CD_N)
#define LDS_BS (LDS_LK + 176)          /* [11][13][18]: B_k x (pose_{k-1}, pose_k, pose_{k+1}) from the IMU factors */
#define LDS_BP (LDS_BS + 11 * 13 * 18)  /* [13][80]: prior rows of the frame whose speed/leg-bias it touches */
#define LDS_S (LDS_BP + 13 * 80)
#define LDS_S_SIZE 4096          /* union: M_k / G_k / T_A of the bias chain  |  diagonal / panel tiles of the Cholesky  |  back-substitution blocks */
#define LDS_RED (LDS_S + LDS_S_SIZE)
#define LDS_COL (LDS_RED + SOLVE_THREADS)
#define LDS_TOTAL (LDS_COL + 176)

extern "C" size_t vilo_solve_lds_bytes() { return (size_t)LDS_TOTAL * sizeof(double); }

// 13x13 Cholesky by one wave: lane i (< 13) owns row i in registers, pivots broadcast with shuffles.
// A: LDS 13x13 row-major in, L (lower, zeros above) written to Lout. Returns 0 ok / 1 not positive definite.
__device__ int chol13_wave(const double *A, double *Lout, double *rinv_out) {
  const int lane = threadIdx.x & 63;
  const int row = lane < 13 ? lane : 0;
  double a[13], l[13];
#pragma unroll
  for (int j = 0; j < 13; ++j) { a[j] = A[row * 13 + j]; l[j] = 0.0; }
  int fail = 0;
#pragma unroll
  for (int j = 0; j < 13; ++j) {
    double s = a[j];
#pragma unroll
    for (int q = 0; q < j; ++q) s -= l[q] * readlane_d(l[q], j);
    double piv = readlane_d(s, j);
    if (!(piv > 0.0) || !isfinite(piv)) { fail = 1; piv = 1.0; }
    const double rinv = rsqrt(piv), ljj = piv * rinv;
    l[j] = (lane == j) ? ljj : (lane > j ? s * rinv : 0.0);
    if (lane == j) rinv_out[j] = rinv;
  }
  if (lane < 13) {
#pragma unroll
    for (int j = 0; j < 13; ++j) Lout[lane * 13 + j] = l[j];
  }
  return fail;
}

// 16x16 Cholesky + inverse of the factor by one wave (diagonal tile of the blocked 80x80 factorisation).
// A: LDS 16x17 row-major in. Lane i (< 16) owns row i of L in registers; pivots broadcast with v_readlane.
// Writes L (lower, incl. diagonal, zeros above) to Ldst (leading dimension ldl) and L^-1 (lower) to Linv (16x17).
// Returns 0 ok / 1 not positive definite (pivot index + 1 in *bad when given).
__device__ int chol16_wave(const double *A, double *Ldst, int ldl, double *Linv) {
  const int lane = threadIdx.x & 63;
  const int row = lane & 15;
  double a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = A[row * 17 + j];
  int fail = 0;
  double myrinv = 1.0;
  // right-looking: after column j is scaled, the updates of the remaining columns are independent FMAs
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
    for (int j = 0; j < 16; ++j) Ldst[lane * ldl + j] = a[j];
    Linv[16 * 17 + lane] = myrinv;   // (scratch row behind the 16 x 17 block)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // column c = lane of L^-1 by forward substitution; L is read back (broadcast) into registers before the dependent chain
  double Lr[120], rv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    rv[i] = Linv[16 * 17 + i];
#pragma unroll
    for (int q = 0; q < i; ++q) Lr[(i * (i - 1)) / 2 + q] = Ldst[i * ldl + q];
  }
  double cl[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    double v = (i == row) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < i; ++q) v -= Lr[(i * (i - 1)) / 2 + q] * cl[q];
    cl[i] = v * rv[i];
  }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) Linv[i * 17 + lane] = (i >= lane) ? cl[i] : 0.0;
  }
  return fail;
}

// The symmetric 80 x 80 pose system as 15 lower 16 x 16 tiles (I >= J) held in FP64-MFMA accumulators: wave w owns
// tiles w, w + 4, w + 8, w + 12 of this list; register r of lane l of a tile is element (16 I + l / 16 + 4 r, 16 J + l % 16).
// The tile list is compile-time per wave (WAVE_DISPATCH instantiates each tile phase once per wave index), so operand
// selection and accumulator indexing are static and the accumulators stay in registers across the phases.
__device__ constexpr int c_tileI[16] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 0};
__device__ constexpr int c_tileJ[16] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 0};
template <int N> struct IC { static constexpr int value = N; };
#define WAVE_DISPATCH(fn) do { if (wv == 0) fn(IC<0>{}); else if (wv == 1) fn(IC<1>{}); else if (wv == 2) fn(IC<2>{}); else fn(IC<3>{}); } while (0)
#define NTILE(WV) ((WV) == 3 ? 3 : 4)
#define WAVE_DISPATCH3(fn) do { if (wv == 0) fn(IC<0>{}); else if (wv == 1) fn(IC<1>{}); else if (wv == 2) fn(IC<2>{}); } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// k_build_solve is one workgroup per window; its phases are separate __noinline__ functions that communicate through
// LDS only (the matrix image in the dynamic LDS block, scalars and tables in KCtx). Keeping the phases apart bounds the
// live ranges: as one function the kernel spilled ~200 doubles per lane to scratch and every small loop paid for it.
// ---------------------------------------------------------------------------------------------------------------------
struct KCtx {
  double *x, *xc, *Tm, *Lkm, *cam_g, *cam_dh2, *cam_y, *cam_scale, *lm_E, *lm_g, *lm_dh2, *lm_scale, *lm_einv, *lm_y;
  const double *wl, *igram, *gs, *pd, *pb0, *phd, *Hp;
  const int *pmap;
  const ChunkMeta *chunks;
  SolverState *st;
  int win, F, L, pn, kb, n_chunks, n_gram, const_mask, gram_off;
  int chain_k;   // lowest frame whose M_k / G_k the bias chain has published (producer: wave 3, consumers: waves 0..2)
  double mu, gnorm2, gmax, qq, gnnorm2, gy;
  SolveParams sp;
  short inv_pmap[CD_N];
  unsigned chunk_tab[64];
  int s_flag[4];
};
extern __shared__ __attribute__((aligned(16))) double lds[];
__shared__ KCtx kc;

#define KB_LOCALS                                                                                                         \
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;                                                              \
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = tid & 15, lk = (tid >> 4) & 3;                             \
  (void)ty; (void)tx; (void)wv; (void)lr; (void)lk;                                                                       \
  double *C = lds + LDS_C, *Ad = lds + LDS_AD, *Ao = lds + LDS_AO, *g = lds + LDS_G, *dh2 = lds + LDS_DH2, *y = lds + LDS_Y; \
  double *tmp = lds + LDS_TMP, *act = lds + LDS_ACT, *Lk = lds + LDS_LK, *S = lds + LDS_S, *red = lds + LDS_RED, *col = lds + LDS_COL; \
  double *Bs = lds + LDS_BS, *Bp = lds + LDS_BP;                                                                          \
  (void)C; (void)Ad; (void)Ao; (void)g; (void)dh2; (void)y; (void)tmp; (void)act; (void)Lk; (void)S; (void)red; (void)col; (void)Bs; (void)Bp; \
  SolverState &st = *kc.st;                                                                                               \
  const SolveParams &sp = kc.sp;                                                                                          \
  const int win = kc.win, F = kc.F, L = kc.L, pn = kc.pn, kb = kc.kb;                                                     \
  (void)win; (void)F; (void)L; (void)pn; (void)kb; (void)sp;                                                              \
  short *inv_pmap = kc.inv_pmap; unsigned *chunk_tab = kc.chunk_tab; int *s_flag = kc.s_flag;                             \
  (void)inv_pmap; (void)chunk_tab; (void)s_flag;                                                                          \
  auto Bval = [&](int k, int i, int p) -> double {                                                                        \
    if (act[CD_B0 + 13 * k + i] == 0.0 || act[p] == 0.0) return 0.0;                                                      \
    double v = 0.0;                                                                                                       \
    if (p < 66) {                                                                                                         \
      const int f = p / 6, df = f - k + 1;                                                                                \
      if (df >= 0 && df <= 2) v = Bs[(k * 13 + i) * 18 + 6 * df + (p - 6 * f)];                                           \
    }                                                                                                                     \
    if (k == kb) v += Bp[i * 80 + p];                                                                                     \
    return v;                                                                                                             \
  };                                                                                                                      \
  (void)Bval;

__device__ __noinline__ void ph_tables() {
  KB_LOCALS
  const WinMeta wm_c = {kc.F, kc.L, kc.n_chunks, 0, 0, 0, kc.const_mask, kc.pn, kc.gram_off, kc.n_gram, 0, kc.kb, 0, 0};
  const WinMeta &wm = wm_c; const int *pmap = kc.pmap; const ChunkMeta *chunkp = kc.chunks;
  // ---- tables ----
  for (int cd = tid; cd < CD_N; cd += SOLVE_THREADS) {
    double a = 1.0;
    if (cd >= CD_EX0 && cd < CD_TD && (wm.const_mask & CONST_EX)) a = 0.0;
    if (cd == CD_TD && (wm.const_mask & CONST_TD)) a = 0.0;
    if (cd == 79 || cd >= CD_B0 + 143) a = 0.0;
    if (cd < 66 && cd / 6 >= F) a = 0.0;
    if (cd >= CD_B0 && cd < CD_B0 + 143) {
      const int k = (cd - CD_B0) / 13, c = (cd - CD_B0) % 13;
      if (k >= F) a = 0.0;
      if (c >= 9 && (wm.const_mask & CONST_LB)) a = 0.0;
    }
    act[cd] = a;
    inv_pmap[cd] = -1;
  }
  // chunk table (s, kmax, first Gram slot) of the window
  if (tid < wm.n_chunks && tid < 64) {
    const ChunkMeta cm = chunkp[tid];
    chunk_tab[tid] = (unsigned)cm.s | ((unsigned)cm.kmax << 8) | ((unsigned)(cm.gram_off - wm.gram_off) << 16);
  }
  __syncthreads();
  for (int i = tid; i < pn; i += SOLVE_THREADS) inv_pmap[pmap[i]] = (short)i;
  __syncthreads();

}

__device__ __noinline__ void ph_assemble() {
  KB_LOCALS
  const double *igram = kc.igram; (void)igram;
  // ---- assembly of the window's normal equations in LDS (no atomics: every target has one owner thread) ----
  // start from the prior's pre-assembled image (all zeros without a prior): coalesced copies instead of zero fill + scatter
  {
    const double *pd = kc.pd;
    // C and Ad are contiguous in LDS and in the image: 8339 doubles = 33 per thread, all loads in flight at once
    double pv[33];
#pragma unroll
    for (int u = 0; u < 33; ++u) { const int e = tid + SOLVE_THREADS * u; pv[u] = pd[min(e, PD_BP - 1)]; }
    double bpv[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) { const int e = tid + SOLVE_THREADS * u; bpv[u] = pd[PD_BP + min(e, 13 * 80 - 1)]; }
#pragma unroll
    for (int u = 0; u < 33; ++u) { const int e = tid + SOLVE_THREADS * u; if (e < PD_BP) C[e] = pv[u]; }
#pragma unroll
    for (int u = 0; u < 5; ++u) { const int e = tid + SOLVE_THREADS * u; if (e < 13 * 80) Bp[e] = bpv[u]; }
    for (int e = tid; e < 10 * 169; e += SOLVE_THREADS) Ao[e] = 0.0;
  }
  // gradient starts from the prior's b0 + H dx (H dx at the current point was formed by k_accept when it evaluated this
  // point's cost), gathered through the inverse prior map so that every g entry has one writer
  for (int e = tid; e < CD_N; e += SOLVE_THREADS) {
    const int pi = inv_pmap[e];
    g[e] = (pn > 0 && pi >= 0) ? kc.pb0[pi] + kc.phd[pi] : 0.0;
  }
  __syncthreads();
  if (tid == 0) st.phase_clk[2] = clock64();
  {
    // Plain (non-atomic) read-modify-write scatter: the work is split so that every target of C / g / Ad / Ao has exactly
    // one owner thread. Two packed Gram entries can hit the same target only if they are "twins" (the same local pair
    // taken once in the pose_s block and once in the pose_j block; for IMU factors once in the frame-i half and once in
    // the frame-j half of the previous factor), so a thread owns an entry together with its twin.
    auto rmw = [&](double *base, int hi, int lo, double v) {   // lower position + mirror inside a diagonal 16-block
      base[hi * CLD + lo] += v;
      if (hi != lo && (hi >> 4) == (lo >> 4)) base[lo * CLD + hi] += v;
    };
    const double *gs = kc.gs;
    const int ns = kc.n_gram;
    // ---- IMU loads first (consumed after the visual walk): groups tid and tid + 256 of the 336 ----
    // classes: I1 pose_i x pose_i (21, twin +19), I2 bias_i x bias_i (91, twin +19), I3 gradient (19, twin +19),
    //          I4 pose_i x pose_j (36), I5 bias_i x bias_j (169)
    int ia[2], ib[2], icls[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int gid = tid + SOLVE_THREADS * h, a = 0, bc = 0, cls = 0;
      if (gid < 21) { cls = 1; int rem = gid; while (rem >= 6 - a) { rem -= 6 - a; ++a; } bc = a + rem; }
      else if (gid < 112) { cls = 2; int rem = gid - 21; while (rem >= 13 - a) { rem -= 13 - a; ++a; } bc = 6 + a + rem; a += 6; }
      else if (gid < 131) { cls = 3; a = gid - 112; bc = 38; }
      else if (gid < 167) { cls = 4; a = (gid - 131) / 6; bc = 19 + (gid - 131) % 6; }
      else if (gid < 336) { cls = 5; a = 6 + (gid - 167) / 13; bc = 25 + (gid - 167) % 13; }
      ia[h] = a; ib[h] = bc; icls[h] = cls;
    }
    // ---- visual Gram slots: 246 owner groups, one per thread ----
    // V1 pose_s x pose_s (21, twin pose_j x pose_j), V2 pose_s x pose_j (36), V3 pose_s x rest (84, twin pose_j x rest),
    // V4 rest x rest (105); rest = ex0 (6) ex1 (6) td r = local columns 12 .. 25
    {
      int a = 0, bc = 0, cls = 0;
      if (tid < 21) { cls = 1; int rem = tid; while (rem >= 6 - a) { rem -= 6 - a; ++a; } bc = a + rem; }
      else if (tid < 57) { cls = 2; a = (tid - 21) / 6; bc = 6 + (tid - 21) % 6; }
      else if (tid < 141) { cls = 3; a = (tid - 57) / 14; bc = 12 + (tid - 57) % 14; }
      else if (tid < 246) { cls = 4; int rem = tid - 141; while (rem >= 14 - a) { rem -= 14 - a; ++a; } bc = 12 + a + rem; a += 12; }
      const bool isg = (bc == 25), dead = (cls == 0) || (a == 25);
      const int e1 = tri26(min(a, 25), bc), e2 = (cls == 1) ? tri26(a + 6, bc + 6) : (cls == 3 ? tri26(a + 6, bc) : e1);
      auto restcd = [](int c) { return c < 18 ? CD_EX0 + c - 12 : (c < 24 ? CD_EX1 + c - 18 : CD_TD); };
      const int rb = (bc >= 12 && bc < 25) ? restcd(bc) : 0, ra_ = (a >= 12 && a < 25) ? restcd(a) : 0;
      // Walk the window's chunks (fixed s, t = 0 .. kmax-1), two per trip = 44 loads in flight. Per chunk:
      //  stage 1: the entry's own target depends on s only (V1, V3, V4): sum over t in a register, one read-modify-write
      //  stage 2: the j-dependent target (twin of V1 / V3, the entry itself for V2): the kmax - 1 targets of a chunk are
      //           distinct, so their reads are batched before their writes (no dependent LDS round trip per slot)
      const int nch = min(kc.n_chunks, 64);
      double *const gb = g;
      for (int ch0 = 0; ch0 < nch; ch0 += 2) {
        double v1[2][11], v2[2][11];
        int cs2[2], km2[2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const unsigned ct = chunk_tab[min(ch0 + c2, nch - 1)];
          cs2[c2] = ct & 255; km2[c2] = (ct >> 8) & 255;
          const int sl0 = ct >> 16;
#pragma unroll
          for (int t = 0; t < 11; ++t) {
            const int tc = min(t, km2[c2] - 1);
            v1[c2][t] = gs[(size_t)(sl0 + tc) * VILO_GRAM + e1];
            v2[c2][t] = gs[(size_t)(sl0 + tc) * VILO_GRAM + e2];
          }
        }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          if (ch0 + c2 >= nch) continue;
          const int s_ = cs2[c2], km = km2[c2];
          // stage 1
          if (cls != 2 && !dead) {
            double sum = 0.0;
#pragma unroll
            for (int t = 0; t < 11; ++t) sum += (t < km) ? v1[c2][t] : 0.0;
            if (isg) gb[cls == 4 ? ra_ : 6 * s_ + a] += sum;
            else if (cls == 1) rmw(C, 6 * s_ + bc, 6 * s_ + a, sum);
            else if (cls == 3) rmw(C, rb, 6 * s_ + a, sum);
            else rmw(C, rb, ra_, sum);
          }
          // stage 2
          if (cls >= 1 && cls <= 3) {
            double *pt[11], *pm[11];
            double o1[11], o2[11];
#pragma unroll
            for (int t = 1; t < 11; ++t) {
              const int j_ = s_ + t;
              int hi, lo;
              if (cls == 1) { hi = 6 * j_ + bc; lo = 6 * j_ + a; }
              else if (cls == 2) { hi = 6 * j_ + (bc - 6); lo = 6 * s_ + a; }
              else { hi = rb; lo = 6 * j_ + a; }
              const bool ong = (cls == 3) && isg;
              pt[t] = ong ? &gb[6 * j_ + a] : &C[hi * CLD + lo];
              pm[t] = (!ong && hi != lo && (hi >> 4) == (lo >> 4)) ? &C[lo * CLD + hi] : nullptr;
            }
#pragma unroll
            for (int t = 1; t < 11; ++t) {
              if (t < km) { o1[t] = *pt[t]; if (pm[t]) o2[t] = *pm[t]; }
            }
#pragma unroll
            for (int t = 1; t < 11; ++t) {
              if (t < km) {
                const double val = (cls == 2) ? v1[c2][t] : v2[c2][t];
                *pt[t] = o1[t] + val;
                if (pm[t]) *pm[t] = o2[t] + val;
              }
            }
          }
        }
      }
    }
    if (tid == 0) st.phase_clk[13] = clock64();
    double vim[2][10], vit[2][10];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int a = ia[h], bc = ib[h], cls = icls[h];
      const int e1 = tri39(a, bc), e2 = (cls >= 1 && cls <= 3) ? tri39(a + 19, cls == 3 ? 38 : bc + 19) : e1;
#pragma unroll
      for (int k = 0; k < 10; ++k) { vim[h][k] = igram[min(k, F - 2) * 780 + e1]; vit[h][k] = igram[min(k, F - 2) * 780 + e2]; }
    }
    __syncthreads();   // the IMU owners below are different threads
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int a = ia[h], bc = ib[h], cls = icls[h];
      if (cls == 0) continue;
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        if (k >= F - 1) continue;
        const double vm = vim[h][k], vt = vit[h][k];
        if (cls == 1) { rmw(C, 6 * k + bc, 6 * k + a, vm); rmw(C, 6 * (k + 1) + bc, 6 * (k + 1) + a, vt); }
        else if (cls == 2) {
          const int ra = a - 6, rc = bc - 6;
          Ad[k * 169 + ra * 13 + rc] += vm; Ad[(k + 1) * 169 + ra * 13 + rc] += vt;
          if (ra != rc) { Ad[k * 169 + rc * 13 + ra] += vm; Ad[(k + 1) * 169 + rc * 13 + ra] += vt; }
        } else if (cls == 3) {
          const int c0 = a < 6 ? a : CD_B0 + (a - 6), ck = a < 6 ? 6 : 13;
          g[c0 + ck * k] += vm; g[c0 + ck * (k + 1)] += vt;
        } else if (cls == 4) rmw(C, 6 * (k + 1) + (bc - 19), 6 * k + a, vm);
        else Ao[k * 169 + (bc - 25) * 13 + (a - 6)] += vm;   // rows frame k+1, cols frame k
      }
    }
    if (tid == 0) st.phase_clk[14] = clock64();
    if (tid == 0) st.phase_clk[15] = clock64();
    if (tid < 13 * 18) {
      const int i = tid / 18, sl = tid % 18, df = sl / 6, c = sl % 6;
      // entry (frame k, row i, pose slot df): from factor k (pose_k / pose_{k+1} columns x bias_k rows) and from factor
      // k - 1 (pose_{k-1} / pose_k columns x bias_k rows); packed index per source fixed per thread, factor walked
      const int e1 = (df == 1) ? tri39(c, 6 + i) : tri39(6 + i, 19 + c);     // factor k     (df = 1: f = k, df = 2: f = k + 1)
      const int e2 = (df == 0) ? tri39(c, 25 + i) : tri39(19 + c, 25 + i);   // factor k - 1 (df = 0: f = k - 1, df = 1: f = k)
      double v1[11], v2[11];
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        v1[k] = igram[min(max(k, 0), F - 2) * 780 + e1];
        v2[k] = igram[min(max(k - 1, 0), F - 2) * 780 + e2];
      }
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const int f = k + df - 1;
        const bool ok = (k < F) && (f >= 0) && (f < F);
        const bool u1 = ok && (k + 1 < F) && (df >= 1);
        const bool u2 = ok && (k > 0) && (df <= 1);
        Bs[(k * 13 + i) * 18 + sl] = (u1 ? v1[k] : 0.0) + (u2 ? v2[k] : 0.0);
      }
    }
  }
  __syncthreads();
  if (tid == 0) st.phase_clk[3] = clock64();
  // constant dims -> identity rows / cols
  if (tid < 80 && act[tid] == 0.0) {   // few dims are inactive (padding, constant extrinsics / td): one thread per such dim
    for (int j = 0; j < 80; ++j) { C[tid * CLD + j] = 0.0; C[j * CLD + tid] = 0.0; }
    C[tid * CLD + tid] = 1.0;
  }
  for (int e = tid; e < 11 * 169; e += SOLVE_THREADS) {
    const int k = e / 169, i = (e % 169) / 13, j = e % 13;
    if (act[CD_B0 + 13 * k + i] == 0.0 || act[CD_B0 + 13 * k + j] == 0.0) Ad[e] = (i == j) ? 1.0 : 0.0;
  }
  for (int e = tid; e < 10 * 169; e += SOLVE_THREADS) {
    const int k = e / 169, i = (e % 169) / 13, j = e % 13;
    if (act[CD_B0 + 13 * (k + 1) + i] == 0.0 || act[CD_B0 + 13 * k + j] == 0.0) Ao[e] = 0.0;
  }
  for (int e = tid; e < CD_N; e += SOLVE_THREADS)
    if (act[e] == 0.0) g[e] = 0.0;
  if (tid == 0) st.phase_clk[21] = clock64();
  if ((tid & 63) == 0 && tid > 0) st.phase_clk[32 + (tid >> 6)] = clock64();
  __syncthreads();
  if (tid == 0) st.phase_clk[22] = clock64();
}

// Block-tridiagonal Cholesky chain of the speed/leg-bias part (wave 3 only, concurrent with the landmark Schur pass).
__device__ __noinline__ void ph_bias_chain() {
  KB_LOCALS
  double *Tm = kc.Tm, *Lkm = kc.Lkm;
  double *Mk = S, *Gk = S + 1859, *TA1 = S + 3718, *rinvk = S + 3887, *TA0 = col;
  {
    // ---- block-tridiagonal Cholesky chain of the speed/leg-bias part (13x13 blocks, frames F-1 .. 0), one wave, no
    //      workgroup barriers:  S_k = A_kk - T_A(k+1)^T T_A(k+1),  L_k = chol(S_k),  M_k = L_k^-1,
    //      T_A(k) = M_k A_{k,k-1},  G_k = M_k T_A(k+1)^T.  Three groups of 13 lanes run the same forward substitution on
    //      different right-hand sides (columns of A_{k,k-1}, of I and of T_A(k+1)^T).
    const int lane = tid & 63, grp = lane >> 4, c = lane & 15;
    const int row = c < 13 ? c : 0;
    double *TAcur = TA0, *TAprev = TA1;
    for (int k = F - 1; k >= 0; --k) {
      if (k == 5 && lane == 0) st.phase_clk[36] = clock64();
      // right-looking 13x13 Cholesky: lane i (of every 16-lane group) keeps row i of S_k; after step j the updates of the
      // remaining columns are independent FMAs; column j of L is broadcast with v_readlane
      double a[13], l[13];
#pragma unroll
      for (int j = 0; j < 13; ++j) { a[j] = Ad[k * 169 + row * 13 + j]; l[j] = 0.0; }
      double myrinv = 1.0;
      int f13 = 0;
#pragma unroll
      for (int j = 0; j < 13; ++j) {
        double piv = readlane_d(a[j], j);
        if (!(piv > 0.0) || !isfinite(piv)) { f13 = 1; piv = 1.0; }
        const double rinv = rsqrt(piv);
        const double lj = (c == j) ? piv * rinv : (c > j ? a[j] * rinv : 0.0);
        l[j] = lj;
        if (c == j) myrinv = rinv;
#pragma unroll
        for (int q = j + 1; q < 13; ++q) a[q] -= lj * readlane_d(lj, q);
      }
      if (f13 && lane == 0) { s_flag[0] = 1; st.pad[0] = 100 + k; }
      if (k == 5 && lane == 0) st.phase_clk[37] = clock64();
      // forward substitutions L x = rhs: T_A(k) columns (group 0), L^-1 columns (group 1), G_k columns (group 2). L goes
      // through LDS once and is read back (broadcast) into registers BEFORE the dependent chain, so the 78 dependent FMAs
      // of a lane wait on nothing but each other
      if (lane < 13) {
#pragma unroll
        for (int j = 0; j < 13; ++j) Lk[lane * 13 + j] = l[j];
        rinvk[lane] = myrinv;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double Lr[78], rv[13], rhs[13];
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        rv[i] = rinvk[i];
#pragma unroll
        for (int q = 0; q < i; ++q) Lr[(i * (i - 1)) / 2 + q] = Lk[i * 13 + q];
        if (grp == 0) rhs[i] = (k > 0) ? Ao[max(k - 1, 0) * 169 + i * 13 + row] : 0.0;
        else if (grp == 1) rhs[i] = (i == c) ? 1.0 : 0.0;
        else rhs[i] = (k < F - 1) ? TAprev[row * 13 + i] : 0.0;
      }
      double cl[13];
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        double v = rhs[i];
#pragma unroll
        for (int q = 0; q < i; ++q) v -= Lr[(i * (i - 1)) / 2 + q] * cl[q];
        cl[i] = v * rv[i];
      }
      if (c < 13 && grp < 3) {
        if (grp == 0) {
#pragma unroll
          for (int i = 0; i < 13; ++i) { TAcur[i * 13 + c] = cl[i]; Tm[k * 13 * 96 + i * 96 + c] = cl[i]; }
        } else if (grp == 1) {
#pragma unroll
          for (int i = 0; i < 13; ++i) { Mk[k * 169 + i * 13 + c] = cl[i]; Lkm[k * 169 + i * 13 + c] = cl[i]; }
        } else {
#pragma unroll
          for (int i = 0; i < 13; ++i) Gk[k * 169 + i * 13 + c] = cl[i];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) *(volatile int *)&kc.chain_k = k;   // M_k / G_k of frame k are visible: the T recurrence may consume them
      if (k == 5 && lane == 0) st.phase_clk[38] = clock64();
      if (k > 0) {
        for (int e = lane; e < 169; e += 64) {
          const int i = e / 13, j = e - 13 * i;
          double sacc = 0.0;
#pragma unroll
          for (int q = 0; q < 13; ++q) sacc += TAcur[q * 13 + i] * TAcur[q * 13 + j];
          Ad[(k - 1) * 169 + e] -= sacc;
        }
      }
      double *sw = TAcur; TAcur = TAprev; TAprev = sw;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (k == 5 && lane == 0) st.phase_clk[39] = clock64();
    }
  }
}

__device__ __forceinline__ int ph_scale_schur_chain() {
  KB_LOCALS
  const double mu = kc.mu;
  double *cam_scale = kc.cam_scale, *Tm = kc.Tm, *Lkm = kc.Lkm;
  const double *wl = kc.wl;
  double *lm_E_ = kc.lm_E, *lm_g_ = kc.lm_g, *lm_dh2_ = kc.lm_dh2, *lm_scale_ = kc.lm_scale, *lm_einv_ = kc.lm_einv, *lm_y_ = kc.lm_y;
  // During q and the landmark Schur pass the 15 lower tiles belong to waves 0..2 (tile t -> wave t % 3, 5 each) while
  // wave 3 runs the block-tridiagonal Cholesky chain of the speed/leg-bias part; afterwards they are redistributed over
  // all four waves (tile t -> wave t % 4) through LDS.
  mfma_d4 acc3[5];
#pragma unroll
  for (int sl = 0; sl < 5; ++sl) acc3[sl] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  auto tile_load3 = [&](auto W_) {
    constexpr int WV = decltype(W_)::value;
#pragma unroll
    for (int sl = 0; sl < 5; ++sl) {
      const int I = c_tileI[WV + 3 * sl], J = c_tileJ[WV + 3 * sl];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc3[sl][r] = C[(16 * I + lk + 4 * r) * CLD + 16 * J + lr];
    }
  };
  WAVE_DISPATCH3(tile_load3);
  if (tid == 0) st.phase_clk[23] = clock64();
  if (tid < 80) y[tid] = C[tid * CLD + tid];
  __syncthreads();
  if (tid == 0) st.phase_clk[4] = clock64();
  // ---- P5: Jacobi scaling (first linearisation), dogleg diagonal, v = D^-2 g ----
  for (int cd = tid; cd < CD_N; cd += SOLVE_THREADS) {
    double hii = 1.0;
    if (cd < CD_B0) hii = y[cd];
    else if (cd < CD_B0 + 143) hii = Ad[((cd - CD_B0) / 13) * 169 + ((cd - CD_B0) % 13) * 14];
    if (act[cd] != 0.0) {
      double sc;
      if (!st.scale_ready) { sc = sp.jacobi_scaling ? 1.0 / (1.0 + sqrt(hii)) : 1.0; cam_scale[cd] = sc; }
      else sc = cam_scale[cd];
      const double d2 = fmin(fmax(sc * sc * hii, sp.min_lm_diagonal), sp.max_lm_diagonal);
      dh2[cd] = d2 / (sc * sc);
      tmp[cd] = g[cd] / dh2[cd];
    } else {
      dh2[cd] = 1.0;
      tmp[cd] = 0.0;
    }
  }
  __syncthreads();
  if (tid == 0) st.phase_clk[27] = clock64();
  // camera part of |D^-1 g|^2, max|g|, q = v^T H v (before regularisation / Schur)
  double part_gn = 0.0, part_q = 0.0, part_gmax = 0.0;
  for (int cd = tid; cd < CD_N; cd += SOLVE_THREADS) {
    part_gn += g[cd] * tmp[cd];
    part_gmax = fmax(part_gmax, fabs(g[cd]));
  }
  auto tile_q = [&](auto W_) {
    constexpr int WV = decltype(W_)::value;
#pragma unroll
    for (int sl = 0; sl < 5; ++sl) {
      const int I = c_tileI[WV + 3 * sl], J = c_tileJ[WV + 3 * sl];
      const double vc = tmp[16 * J + lr], sym = (I == J) ? 1.0 : 2.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) part_q += sym * tmp[16 * I + lk + 4 * r] * acc3[sl][r] * vc;
    }
  };
  WAVE_DISPATCH3(tile_q);
  if (tid == 0) st.phase_clk[28] = clock64();
  if (tid >= 96 && tid < 96 + 143) {
    const int e = tid - 96, k = e / 13, i = e % 13;
    if (k < F) {
      const double vi = tmp[CD_B0 + e];
      double sacc = 0.0;
      for (int j = 0; j < 13; ++j) sacc += Ad[k * 169 + i * 13 + j] * tmp[CD_B0 + 13 * k + j];
      double cross = 0.0;
      if (k > 0)
        for (int j = 0; j < 13; ++j) cross += Ao[(k - 1) * 169 + i * 13 + j] * tmp[CD_B0 + 13 * (k - 1) + j];
      // coupling rows: the IMU part spans poses k-1 .. k+1 (Bs), the prior part (frame kb only) is a plain dot product
      double bp = 0.0;
      if (act[CD_B0 + 13 * k + i] != 0.0) {
#pragma unroll
        for (int sl = 0; sl < 18; ++sl) {
          const int f = k + sl / 6 - 1, pc = min(max(6 * f + sl % 6, 0), 65);
          bp += (f >= 0 && f < F) ? Bs[(k * 13 + i) * 18 + sl] * act[pc] * tmp[pc] : 0.0;
        }
        if (k == kb) {
#pragma unroll 16
          for (int p = 0; p < 80; ++p) bp += Bp[i * 80 + p] * act[p] * tmp[p];
        }
      }
      part_q += vi * (sacc + 2.0 * cross + 2.0 * bp);
    }
  }
  if (tid == 96) st.phase_clk[29] = clock64();
  // ---- P6: landmarks pass 1 ----
  double *lm_E = lm_E_, *lm_g = lm_g_, *lm_dh2 = lm_dh2_, *lm_scale = lm_scale_, *lm_einv = lm_einv_, *lm_y = lm_y_;
  for (int l = tid; l < L; l += SOLVE_THREADS) {
    const double E = lm_E[l], gl = lm_g[l];
    double sc;
    if (!st.scale_ready) { sc = sp.jacobi_scaling ? 1.0 / (1.0 + sqrt(E)) : 1.0; lm_scale[l] = sc; }
    else sc = lm_scale[l];
    const double d2 = fmin(fmax(sc * sc * E, sp.min_lm_diagonal), sp.max_lm_diagonal) / (sc * sc);
    lm_dh2[l] = d2;
    const double vl = gl / d2;
    part_q += E * vl * vl;   // the cross term 2 vl w_l^T v is accumulated in the Schur pass below
    lm_y[l] = vl;            // (scratch until the back-substitution overwrites it)
    part_gn += gl * vl;
    part_gmax = fmax(part_gmax, fabs(gl));
    lm_einv[l] = 1.0 / (E + mu * d2);
  }
  if (tid == 0) st.phase_clk[30] = clock64();
  const double gnorm2 = block_sum(part_gn, red);
  const double gmax = block_max(part_gmax, red);
  if (!sp.fixed_iterations && gmax <= sp.gradient_tolerance) {
    if (tid == 0) { st.gmax = gmax; st.done = 1; st.termination = 1; st.step_valid = 0; }
    return 1;
  }
  if (tid == 0) st.phase_clk[5] = clock64();
  // regularise the speed/leg-bias diagonal blocks (the pose part is regularised after the tile redistribution)
  for (int e = tid; e < 11 * 13; e += SOLVE_THREADS) Ad[(e / 13) * 169 + (e % 13) * 14] += mu * dh2[CD_B0 + e];
  if (tid == 0) { s_flag[0] = 0; *(volatile int *)&kc.chain_k = F; }
  __syncthreads();   // also: lm_einv / lm_y written above are read through global memory below
  double *Mk = S, *Gk = S + 1859, *TA1 = S + 3718, *rinvk = S + 3887, *TA0 = col;
  double actv[5], vv[5], yacc0 = 0.0, yacc1 = 0.0, qacc = 0.0;
#pragma unroll
  for (int X = 0; X < 5; ++X) { actv[X] = act[16 * X + lr]; vv[X] = tmp[16 * X + lr]; }
  if (wv == 3) {
    ph_bias_chain();
  } else {
    // ---- Schur complement of the landmarks on the FP64 matrix cores: C -= sum_l w_l w_l^T / (E_l + mu dhat_l^2). One
    //      k-step = 4 landmarks; the operand of tile row X (lane: w[16 X + l % 16][4 kk + l / 16]) serves as A of tiles
    //      (X, .) and as B of tiles (., X): 5 row-coalesced global loads and 5 MFMAs per k-step per wave, no LDS. The same
    //      operands give rhs_P -= sum_l w_l g_l / (...) and the 2 vl w_l^T v term of q.
    const int nks = (L + 3) >> 2;
    auto tile_schur = [&](auto W_) {
      constexpr int WV = decltype(W_)::value;
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
          for (int sl = 0; sl < 5; ++sl) {
            const int I = c_tileI[WV + 3 * sl], J = c_tileJ[WV + 3 * sl];
            acc3[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(-(op[I] * ei), op[J], acc3[sl], 0, 0, 0);
          }
          yacc0 += op[WV] * ge; qacc += op[WV] * vv[WV] * vl;
          if (WV < 2) { yacc1 += op[WV + 3] * ge; qacc += op[WV + 3] * vv[WV + 3] * vl; }
        }
      };
      // 4 k-steps (16 landmarks) per trip, the next trip's 32 loads in flight behind the current trip's 20 MFMAs;
      // landmarks past L are clamped to a valid address and masked through their 1 / (E + mu d2) factor
      ldtrip(0, 0);
      for (int kk0 = 0; kk0 < nks; kk0 += 8) {
        ldtrip(kk0 + 4, 1);
        dotrip(kk0, 0);
        ldtrip(kk0 + 8, 0);
        dotrip(kk0 + 4, 1);
      }
      // hand the tiles to their 4-wave owners through C (lower tile positions)
#pragma unroll
      for (int sl = 0; sl < 5; ++sl) {
        const int I = c_tileI[WV + 3 * sl], J = c_tileJ[WV + 3 * sl];
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(16 * I + lk + 4 * r) * CLD + 16 * J + lr] = acc3[sl][r];
      }
    };
    WAVE_DISPATCH3(tile_schur);
    part_q += 2.0 * qacc;
    yacc0 += __shfl_xor(yacc0, 16, 64); yacc0 += __shfl_xor(yacc0, 32, 64);
    yacc1 += __shfl_xor(yacc1, 16, 64); yacc1 += __shfl_xor(yacc1, 32, 64);
    if (lk == 0) { y[16 * wv + lr] = yacc0; if (wv < 2) y[16 * (wv + 3) + lr] = yacc1; }
    if (tid == 0) st.phase_clk[40] = clock64();
    // ---- T(k) = [T_B(k) | t_g(k)] = M_k [B_k | rhs_k] - G_k T(k+1): the 80 + 1 columns are independent, so each wave
    //      carries its 16-column tiles (wave w < 3: tiles w and w + 3; tile 5 = rhs column) through all frames without any
    //      barrier. The previous result is already in B-operand layout: register kk of a lane is row 4 kk + l / 16. ----
      {
      const int X0 = wv, X1 = wv + 3;
      mfma_d4 t0 = {0.0, 0.0, 0.0, 0.0}, t1 = {0.0, 0.0, 0.0, 0.0};
      for (int k = F - 1; k >= 0; --k) {
        while (*(volatile int *)&kc.chain_k > k) __builtin_amdgcn_s_sleep(4);   // frame k published by the chain (wave 3)
        asm volatile("" ::: "memory");
        mfma_d4 n0 = {0.0, 0.0, 0.0, 0.0}, n1 = {0.0, 0.0, 0.0, 0.0};
        double am[4], ag[4];
  #pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int q = 4 * kk + lk;
          const bool in = (lr < 13) && (q < 13);
          const int idx = k * 169 + min(lr, 12) * 13 + min(q, 12);
          const double m = Mk[idx], gg = Gk[idx];
          am[kk] = in ? m : 0.0;
          ag[kk] = (in && k < F - 1) ? -gg : 0.0;
        }
  #pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int q = 4 * kk + lk, qc = min(q, 12);
          double b0 = Bval(k, qc, 16 * X0 + lr);
          b0 = (q < 13) ? b0 : 0.0;
          n0 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[kk], b0, n0, 0, 0, 0);
          if (X1 < 6) {
            double b1 = (X1 < 5) ? Bval(k, qc, 16 * min(X1, 4) + lr) : ((lr == 0) ? g[CD_B0 + 13 * k + qc] : 0.0);
            b1 = (q < 13) ? b1 : 0.0;
            n1 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[kk], b1, n1, 0, 0, 0);
          }
        }
  #pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          n0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[kk], t0[kk], n0, 0, 0, 0);
          if (X1 < 6) n1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[kk], t1[kk], n1, 0, 0, 0);
        }
        t0 = n0; t1 = n1;
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = lk + 4 * r;
          if (row < 13) {
            Tm[k * 13 * 96 + row * 96 + 13 + 16 * X0 + lr] = n0[r];
            if (X1 < 5) Tm[k * 13 * 96 + row * 96 + 13 + 16 * X1 + lr] = n1[r];
            else if (X1 == 5 && lr == 0) Tm[k * 13 * 96 + row * 96 + 93] = n1[r];
          }
        }
      }
    }
  }
  const double qq = block_sum(part_q, red);   // (also the barrier that publishes y, the tiles in C and the chain's output)
  if (tid == 0) { kc.gnorm2 = gnorm2; kc.gmax = gmax; kc.qq = qq; }
  return 0;
}

__device__ __noinline__ void ph_elim_chol() {
  KB_LOCALS
  const double mu = kc.mu;
  double *Tm = kc.Tm;
  double *Mk = S, *Gk = S + 1859;
  if (tid == 0) st.phase_clk[6] = clock64();
  for (int cd = tid; cd < CD_N; cd += SOLVE_THREADS) tmp[cd] = g[cd] - ((cd < 80) ? y[cd] : 0.0);   // reduced rhs
  mfma_d4 acc[4];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) acc[sl] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  auto tile_load = [&](auto W_) {
    constexpr int WV = decltype(W_)::value;
#pragma unroll
    for (int sl = 0; sl < NTILE(WV); ++sl) {
      const int I = c_tileI[WV + 4 * sl], J = c_tileJ[WV + 4 * sl];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[sl][r] = C[(16 * I + lk + 4 * r) * CLD + 16 * J + lr];
        if (I == J && lk + 4 * r == lr) acc[sl][r] += mu * dh2[16 * I + lr];   // regularise: diag += mu dhat^2
      }
    }
  };
  WAVE_DISPATCH(tile_load);
  __syncthreads();   // tmp (reduced rhs) complete
  __syncthreads();   // T in global memory is read by every wave below
  // ---- C -= sum_k T_B(k)^T T_B(k) (rank 143) on the matrix cores, rhs_P -= sum_k T_B(k)^T t_g(k); operands from Tm ----
  {
    double yr0 = 0.0, yr1 = 0.0;
    auto tile_rank = [&](auto W_) {
      constexpr int WV = decltype(W_)::value;
      for (int k = 0; k < F; ++k) {
        double opT[4][5], tg[4];
        const double *Tk = Tm + (size_t)k * 13 * 96;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int qc = min(4 * kk + lk, 12);
          tg[kk] = Tk[qc * 96 + 93];
#pragma unroll
          for (int X = 0; X < 5; ++X) opT[kk][X] = Tk[qc * 96 + 13 + 16 * X + lr];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const bool in = 4 * kk + lk < 13;
#pragma unroll
          for (int X = 0; X < 5; ++X) opT[kk][X] = in ? opT[kk][X] : 0.0;
#pragma unroll
          for (int sl = 0; sl < NTILE(WV); ++sl) {
            const int I = c_tileI[WV + 4 * sl], J = c_tileJ[WV + 4 * sl];
            acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(-opT[kk][I], opT[kk][J], acc[sl], 0, 0, 0);
          }
          yr0 += opT[kk][WV] * tg[kk];
          if (WV == 0) yr1 += opT[kk][4] * tg[kk];
        }
      }
    };
    WAVE_DISPATCH(tile_rank);
    yr0 += __shfl_xor(yr0, 16, 64); yr0 += __shfl_xor(yr0, 32, 64);
    yr1 += __shfl_xor(yr1, 16, 64); yr1 += __shfl_xor(yr1, 32, 64);
    if (lk == 0) { tmp[16 * wv + lr] -= yr0; if (wv == 0) tmp[64 + lr] -= yr1; }
  }
  lds_barrier();
  if (tid == 0) st.phase_clk[7] = clock64();
  // ---- dense Cholesky of the 80x80 reduced pose system: register tiles, pivot column broadcast through LDS,
  //      reciprocal square root of the pivot (one Newton step on v_rsq_f64) instead of sqrt + 80 divisions ----
  // ---- dense Cholesky of the 80x80 reduced pose system, blocked by 16: diagonal tile by one wave (registers +
  //      v_readlane), panel L_Ij = A_Ij L_jj^-T and trailing update A_IJ -= L_Ij L_Jj^T on the FP64 matrix cores;
  //      2 workgroup barriers per block column (10 in total). L is left in C (lower triangle) for the solves ----
  double *P16 = S, *D16 = S + 1100, *LI16 = S + 1400;
  auto tile_chol = [&](auto W_) {
    constexpr int WV = decltype(W_)::value;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
#pragma unroll
      for (int sl = 0; sl < NTILE(WV); ++sl) {
        const int I = c_tileI[WV + 4 * sl], J = c_tileJ[WV + 4 * sl];
        if (I == j && J == j) {
#pragma unroll
          for (int r = 0; r < 4; ++r) D16[(lk + 4 * r) * 17 + lr] = acc[sl][r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const int f16 = chol16_wave(D16, C + (16 * j) * CLD + 16 * j, CLD, LI16);
          if (f16 && (tid & 63) == 0) { if (st.pad[1] == 0) st.pad[1] = 1000 + 16 * j; s_flag[1] = 1; }
        }
      }
      lds_barrier();
#pragma unroll
      for (int sl = 0; sl < NTILE(WV); ++sl) {
        const int I = c_tileI[WV + 4 * sl], J = c_tileJ[WV + 4 * sl];
        if (J == j && I > j) {
          double *Pt = P16 + (I - j - 1) * 272;
#pragma unroll
          for (int r = 0; r < 4; ++r) Pt[(lk + 4 * r) * 17 + lr] = acc[sl][r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          mfma_d4 nacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) nacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pt[lr * 17 + 4 * kk + lk], LI16[lr * 17 + 4 * kk + lk], nacc, 0, 0, 0);
          acc[sl] = nacc;
#pragma unroll
          for (int r = 0; r < 4; ++r) C[(16 * I + lk + 4 * r) * CLD + 16 * j + lr] = nacc[r];
        }
      }
      lds_barrier();
#pragma unroll
      for (int sl = 0; sl < NTILE(WV); ++sl) {
        const int I = c_tileI[WV + 4 * sl], J = c_tileJ[WV + 4 * sl];
        if (J > j) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(-C[(16 * I + lr) * CLD + 16 * j + 4 * kk + lk], C[(16 * J + lr) * CLD + 16 * j + 4 * kk + lk], acc[sl], 0, 0, 0);
        }
      }
    }
  };
  WAVE_DISPATCH(tile_chol);
  lds_barrier();
}

__device__ __forceinline__ void ph_solve() {
  KB_LOCALS
  double *Tm = kc.Tm, *Lkm = kc.Lkm;
  const double *wl = kc.wl;
  double *lm_g = kc.lm_g, *lm_dh2 = kc.lm_dh2, *lm_einv = kc.lm_einv, *lm_y = kc.lm_y;
  double *U = S;               // [11][13]
  double *TA = S + 160;        // [11][169]  T_A blocks
  double *LI = S + 2048;       // [11][169]  L_k^-1 blocks
  if (tid < 80) col[tid] = 1.0 / C[tid * CLD + tid];
  lds_barrier();
  // ---- phase A: wave 0 solves L L^T yP = rhs (lane owns rows lane and lane + 64; pivots by v_readlane; the factor is
  //      read in blocks of 16 columns into registers so that the 160 dependent steps touch no memory); meanwhile waves
  //      1..3 stage T_A / L_k^-1 in LDS and fetch their rows of T_B (consumed in phase B) ----
  double trow[81];
  const int urow = tid - 64;   // (frame, row) of the coupling block handled by this thread in phase B
  if (wv == 0) {
    const int lane = tid;
    double b0 = tmp[lane], b1 = lane < 16 ? tmp[lane + 64] : 0.0;
    const int r1 = min(lane + 64, 79);
#pragma unroll
    for (int jb = 0; jb < 5; ++jb) {
      double l0[16], l1[16], ri[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) { l0[jj] = C[lane * CLD + 16 * jb + jj]; l1[jj] = C[r1 * CLD + 16 * jb + jj]; ri[jj] = col[16 * jb + jj]; }
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
      for (int jj = 0; jj < 16; ++jj) { c0[jj] = C[(16 * jb + jj) * CLD + lane]; c1[jj] = C[(16 * jb + jj) * CLD + r1]; ri[jj] = col[16 * jb + jj]; }
#pragma unroll
      for (int jj = 15; jj >= 0; --jj) {
        const int j = 16 * jb + jj;
        const double yj = readlane_d((jb < 4) ? b0 : b1, j & 63) * ri[jj];
        if (lane < j) b0 -= c0[jj] * yj;
        if (lane < 16 && lane + 64 < j) b1 -= c1[jj] * yj;
        if (lane == (j & 63)) { if (jb < 4) b0 = yj; else b1 = yj; }
      }
    }
    y[lane] = b0;
    if (lane < 16) y[lane + 64] = b1;
  } else {
    // (the elimination's global stores of T / L^-1 were ordered by the barriers at the end of ph_elim_chol)
    for (int e = tid - 64; e < F * 169; e += SOLVE_THREADS - 64) {
      const int k = e / 169, r = (e % 169) / 13, c = e % 13;
      TA[e] = Tm[(size_t)k * 13 * 96 + r * 96 + c];
      LI[e] = Lkm[e];
    }
    if (urow < 13 * F) {
      const double *tr = Tm + (size_t)(urow / 13) * 13 * 96 + (urow % 13) * 96 + 13;
#pragma unroll
      for (int q = 0; q < 81; ++q) trow[q] = tr[q];   // T_B row (80) and t_g (column 93 = tr[80])
    }
  }
  __syncthreads();
  if (tid == 0) st.phase_clk[9] = clock64();
  // ---- phase B: u_k = t_g - T_B yP ----
  if (wv != 0 && urow < 13 * F) {
    double sacc = trow[80];
#pragma unroll
    for (int q = 0; q < 80; ++q) sacc -= trow[q] * y[q];
    U[urow] = sacc;
  }
  __syncthreads();
  // ---- phase C: wave 0 runs the chain over frames y_k = L_k^-T (u_k - T_A y_{k-1}) (one LDS round trip per frame); waves
  //      1..3 back-substitute the landmarks (they only need yP): all 80 coupling entries of a landmark in flight at once ----
  double part_gnn = 0.0, part_gy = 0.0;
  auto landmark = [&](int l) {
    double wcol[80];
#pragma unroll
    for (int a = 0; a < 80; ++a) wcol[a] = wl[(size_t)a * L + l];
    const double gl = lm_g[l], ei = lm_einv[l], d2 = lm_dh2[l];
    double tl = 0.0;
#pragma unroll
    for (int a = 0; a < VILO_NPU; ++a) tl += wcol[a] * act[a] * y[a];
    const double yl = (gl - tl) * ei;
    lm_y[l] = yl;
    part_gnn += d2 * yl * yl;
    part_gy += gl * yl;
  };
  if (wv == 0) {
    const int lane = tid, row = lane < 13 ? lane : 0;
    double *rb = col;   // rhs broadcast buffer
    for (int k = 0; k < F; ++k) {
      double rhs = U[k * 13 + row];
      if (k > 0) {
#pragma unroll
        for (int q = 0; q < 13; ++q) rhs -= TA[k * 169 + row * 13 + q] * y[CD_B0 + 13 * (k - 1) + q];
      }
      if (lane < 13) rb[lane] = rhs;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double yk = 0.0;
#pragma unroll
      for (int q = 0; q < 13; ++q) yk += LI[k * 169 + q * 13 + row] * rb[q];   // (L^-T rhs)_row = sum_q Linv[q][row] rhs[q]
      if (lane < 13) y[CD_B0 + 13 * k + lane] = yk;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int l = 192 + tid; l < L; l += SOLVE_THREADS) landmark(l);
  } else {
    for (int l = tid - 64; l < L; l += SOLVE_THREADS) landmark(l);
  }
  __syncthreads();
  if (tid == 0) st.phase_clk[10] = clock64();
  // ---- camera part of the norms ----
  for (int cd = tid; cd < CD_N; cd += SOLVE_THREADS) {
    if (act[cd] == 0.0) y[cd] = 0.0;
  }
  __syncthreads();
  for (int cd = tid; cd < CD_N; cd += SOLVE_THREADS) {
    part_gnn += dh2[cd] * y[cd] * y[cd] * act[cd];
    part_gy += g[cd] * y[cd];
  }
  const double gnnorm2 = block_sum(part_gnn, red);
  const double gy = block_sum(part_gy, red);
  if (tid == 0) { kc.gnnorm2 = gnnorm2; kc.gy = gy; }
}

__global__ void __launch_bounds__(SOLVE_THREADS) k_build_solve(BatchDev b, SolveParams sp_in) {
  const int win = blockIdx.x;
  SolverState &st = b.st[win];
  if (st.done) return;
  const int tid = threadIdx.x;
  double *g = lds + LDS_G, *dh2 = lds + LDS_DH2, *y = lds + LDS_Y, *tmp = lds + LDS_TMP;
  const WinMeta wm = b.win[win];
  double *x = b.x + (size_t)win * XSTRIDE, *xc = b.xc + (size_t)win * XSTRIDE;
  double *cam_g = b.cam_g + (size_t)win * CD_N, *cam_dh2 = b.cam_dh2 + (size_t)win * CD_N, *cam_y = b.cam_y + (size_t)win * CD_N;
  if (tid == 0) {
    kc.x = x; kc.xc = xc;
    kc.Tm = b.Tm + (size_t)win * 11 * 13 * 96; kc.Lkm = b.Lk + (size_t)win * 11 * 169;
    kc.cam_g = cam_g; kc.cam_dh2 = cam_dh2; kc.cam_y = cam_y; kc.cam_scale = b.cam_scale + (size_t)win * CD_N;
    kc.lm_E = b.lm_E + wm.lm_off; kc.lm_g = b.lm_g + wm.lm_off; kc.lm_dh2 = b.lm_dh2 + wm.lm_off;
    kc.lm_scale = b.lm_scale + wm.lm_off; kc.lm_einv = b.lm_einv + wm.lm_off; kc.lm_y = b.lm_y + wm.lm_off;
    kc.wl = b.lm_w + 80 * (size_t)wm.lm_off; kc.igram = b.imu_gram + (size_t)win * 10 * 780;
    kc.gs = b.gram + (size_t)wm.gram_off * VILO_GRAM; kc.pd = b.prior_dense + (size_t)win * PD_N;
    kc.pb0 = b.prior_b0 + (size_t)win * 96; kc.phd = b.prior_hd + (size_t)win * 96; kc.Hp = b.prior_H + (size_t)win * 96 * 96;
    kc.pmap = b.prior_map + (size_t)win * 96; kc.chunks = b.chunk + wm.chunk_off; kc.st = &st;
    kc.win = win; kc.F = wm.n_frames; kc.L = wm.L; kc.pn = wm.prior_n; kc.kb = wm.pad; kc.n_chunks = wm.n_chunks;
    kc.n_gram = wm.n_gram; kc.const_mask = wm.const_mask; kc.gram_off = wm.gram_off;
    kc.sp = sp_in;
  }
  __syncthreads();
  const SolveParams &sp = kc.sp;

  if (st.need_lin) {
    if (tid == 0) st.phase_clk[0] = clock64();
    ph_tables();
    bool solved = false;
    while (!solved) {
      __syncthreads();
      if (tid == 0) { kc.mu = st.mu; kc.s_flag[0] = 0; kc.s_flag[1] = 0; st.phase_clk[1] = clock64(); }
      __syncthreads();
      ph_assemble();
      if (ph_scale_schur_chain()) return;
      ph_elim_chol();
      const int fail = kc.s_flag[0] | kc.s_flag[1];
      if (fail) {
        // DoglegStrategy::ComputeGaussNewtonStep: mu *= 10 and retry while mu < max_mu (1.0)
        __syncthreads();
        if (tid == 0) { st.mu *= 10.0; }
        __syncthreads();
        if (!(st.mu < 1.0)) {
          if (tid == 0) { st.lin_fail = 1; st.step_valid = 0; st.gnorm2 = kc.gnorm2; st.q = kc.qq; st.gmax = kc.gmax; st.scale_ready = 1; }
          return;
        }
        continue;
      }
      if (tid == 0) st.phase_clk[8] = clock64();
      ph_solve();
      __syncthreads();
      const double gnnorm2 = kc.gnnorm2, gy = kc.gy;
      if (!(isfinite(gnnorm2) && isfinite(gy))) {   // IsArrayValid(gauss_newton_step_) failed
        if (tid == 0) st.mu *= 10.0;
        __syncthreads();
        if (!(st.mu < 1.0)) {
          if (tid == 0) { st.lin_fail = 1; st.step_valid = 0; st.scale_ready = 1; }
          return;
        }
        continue;
      }
      for (int cd = tid; cd < CD_N; cd += SOLVE_THREADS) { cam_g[cd] = g[cd]; cam_dh2[cd] = dh2[cd]; cam_y[cd] = y[cd]; }
      if (tid == 0) {
        st.gnorm2 = kc.gnorm2; st.gnnorm2 = gnnorm2; st.gdotgn = -gy; st.q = kc.qq; st.gmax = kc.gmax;
        st.alpha = kc.gnorm2 / kc.qq;
        st.scale_ready = 1;
        st.lin_fail = 0;
      }
      solved = true;
    }
    __syncthreads();
  } else {
    for (int cd = tid; cd < CD_N; cd += SOLVE_THREADS) { g[cd] = cam_g[cd]; dh2[cd] = cam_dh2[cd]; y[cd] = cam_y[cd]; }
    __syncthreads();
  }

  if (tid == 0) st.phase_clk[11] = clock64();
  // ---- P11: dogleg step for the current radius, candidate camera state ----
  if (tid == 0) {
    if (st.radius <= sp.min_radius) { st.done = 1; st.termination = 1; st.step_valid = 0; }
    else dogleg_scalars(st);
  }
  __syncthreads();
  if (st.done || !st.step_valid) return;
  const double ca = st.coef_a, cb = st.coef_b;
  for (int cd = tid; cd < CD_N; cd += SOLVE_THREADS) tmp[cd] = -ca * g[cd] / dh2[cd] - cb * y[cd];
  __syncthreads();
  if (tid < 11) pose_plus(x + XO_POSE + 7 * tid, tmp + 6 * tid, xc + XO_POSE + 7 * tid);
  else if (tid < 13) pose_plus(x + XO_EX + 7 * (tid - 11), tmp + CD_EX0 + 6 * (tid - 11), xc + XO_EX + 7 * (tid - 11));
  else if (tid == 13) xc[XO_TD] = x[XO_TD] + tmp[CD_TD];
  else if (tid >= 32 && tid < 32 + 143) {
    const int e = tid - 32, k = e / 13, c = e % 13;
    if (c < 9) xc[XO_SB + 9 * k + c] = x[XO_SB + 9 * k + c] + tmp[CD_B0 + e];
    else xc[XO_LB + 4 * k + (c - 9)] = x[XO_LB + 4 * k + (c - 9)] + tmp[CD_B0 + e];
  }
  if (tid == 0) st.phase_clk[12] = clock64();
}

// =================================================================================================
// k_accept: candidate cost, step quality, accept / reject (TrustRegionMinimizer::{IsStepSuccessful,
// HandleSuccessfulStep, HandleUnsuccessfulStep, HandleInvalidStep} + DoglegStrategy::Step{Accepted,Rejected,IsInvalid})
// =================================================================================================
struct AcceptParams {
  double min_relative_decrease, function_tolerance, parameter_tolerance;
  int max_num_iterations, fixed_iterations, init_mode, pad;
};

__global__ void __launch_bounds__(128) k_accept(BatchDev b, AcceptParams ap) {
  __shared__ double red[128];
  __shared__ double dxs[VILO_MAX_PRIOR_DIM];
  __shared__ int accept_s;
  const int win = blockIdx.x, tid = threadIdx.x;
  SolverState &st = b.st[win];
  if (st.done) return;
  const WinMeta wm = b.win[win];
  double *x = b.x + (size_t)win * XSTRIDE, *xc = b.xc + (size_t)win * XSTRIDE;
  if (!ap.init_mode && !st.step_valid) {
    if (tid == 0) {
      // HandleInvalidStep
      st.num_invalid++;
      if (st.num_invalid > 5) { st.done = 1; st.termination = 2; }
      st.mu *= 10.0;
      st.need_lin = 1;
      st.iter++;
      if (st.iter < 64) { st.cost_trace[st.iter] = st.x_cost; st.radius_trace[st.iter] = st.radius; }
      if (st.iter >= ap.max_num_iterations && !st.done) { st.done = 1; st.termination = 0; }
    }
    return;
  }
  // candidate cost = 1/2 (visual rho sums + |imu residuals|^2 + |prior residual|^2)
  double part = 0.0;
  for (int c = tid; c < wm.n_waves * VILO_MAX_FRAMES; c += 128) part += b.chunk_cost[(size_t)wm.wave_off * VILO_MAX_FRAMES + c];
  const double vis = block_sum(part, red);
  part = 0.0;
  for (int k = tid; k + 1 < wm.n_frames; k += 128) part += b.imu_cost[(size_t)win * 10 + k];
  const double imu = block_sum(part, red);
  double pri = 0.0, my_hd = 0.0;   // my_hd: row tid of H dx at the candidate (becomes the gradient term when accepted)
  if (wm.prior_n > 0) {
    const int n = wm.prior_n;
    if (tid < wm.prior_nb)
      prior_dx(xc + b.prior_bstate[win * 40 + tid], b.prior_x0 + (size_t)win * 280 + b.prior_bxoff[win * 40 + tid],
               b.prior_bsize[win * 40 + tid], dxs + b.prior_bidx[win * 40 + tid]);
    __syncthreads();
    const double *Hp = b.prior_H + (size_t)win * 96 * 96, *b0 = b.prior_b0 + (size_t)win * 96;
    part = 0.0;
    if (tid < n) {   // n <= 96 < 128: one row per thread
      double sacc = 0.0;
#pragma unroll 16
      for (int q = 0; q < n; ++q) sacc += Hp[(size_t)q * n + tid] * dxs[q];
      part = dxs[tid] * (sacc + 2.0 * b0[tid]);
      my_hd = sacc;
    }
    pri = block_sum(part, red) + b.prior_c0[win];
  }
  double cand = 0.5 * (vis + imu + pri);
  if (!isfinite(cand)) cand = 1.7976931348623157e308;
  if (ap.init_mode) {
    if (tid == 0) {
      st.x_cost = cand; st.cand_cost = cand; st.vis_cost = vis; st.imu_cost = imu; st.prior_cost = pri;
      st.cost_trace[0] = cand; st.radius_trace[0] = st.radius;
      // a non-finite evaluation at the initial point: ceres::Solve fails in IterationZero ("Residual and Jacobian evaluation
      // failed", ResidualBlock::Evaluate's IsArrayValid) and leaves the parameters alone; this window is done, the others go on
      if (!(cand < 1.7976931348623157e308)) { st.done = 1; st.termination = 2; }
    }
    if (tid < wm.prior_n) b.prior_hd[(size_t)win * 96 + tid] = my_hd;
    return;
  }
  // ambient-space norms for ParameterToleranceReached
  bool converged = false;
  if (!ap.fixed_iterations) {
    double pn = 0.0, ps = 0.0;
    for (int e = tid; e < XSTRIDE; e += 128) {
      bool on = e < XO_TD + 1 && !(e >= XO_EX && e < XO_TD && (wm.const_mask & CONST_EX)) && !(e == XO_TD && (wm.const_mask & CONST_TD)) &&
                !(e >= XO_LB && e < XO_EX && (wm.const_mask & CONST_LB));
      if (on) { pn += x[e] * x[e]; ps += (x[e] - xc[e]) * (x[e] - xc[e]); }
    }
    for (int l = tid; l < wm.L; l += 128) {
      const double a = b.lam[wm.lm_off + l], c = b.lamc[wm.lm_off + l];
      pn += a * a; ps += (a - c) * (a - c);
    }
    const double xn = sqrt(block_sum(pn, red)), sn = sqrt(block_sum(ps, red));
    if (sn <= ap.parameter_tolerance * (xn + ap.parameter_tolerance)) converged = true;
    if (!converged && fabs(st.x_cost - cand) <= ap.function_tolerance * st.x_cost) converged = true;
  }
  if (converged) {
    if (tid == 0) { st.done = 1; st.termination = 1; st.cand_cost = cand; }
    return;
  }
  if (tid == 0) {
    const double rel = (st.x_cost - cand) / st.model_cost_change;
    st.cand_cost = cand;
    st.num_invalid = 0;
    if (rel > ap.min_relative_decrease) {
      accept_s = 1;
      st.x_cost = cand; st.vis_cost = vis; st.imu_cost = imu; st.prior_cost = pri;
      if (rel < 0.25) st.radius *= 0.5;
      if (rel > 0.75) st.radius = fmax(st.radius, 3.0 * st.dogleg_step_norm);
      st.mu = fmax(1e-8, 2.0 * st.mu / 10.0);
      st.need_lin = 1;
      st.num_successful++;
    } else {
      accept_s = 0;
      st.radius *= 0.5;
      st.need_lin = 0;
    }
    st.iter++;
    if (st.iter < 64) { st.cost_trace[st.iter] = st.x_cost; st.radius_trace[st.iter] = st.radius; }
    if (st.iter >= ap.max_num_iterations) { st.done = 1; st.termination = 0; }
  }
  __syncthreads();
  if (accept_s) {
    if (tid < wm.prior_n) b.prior_hd[(size_t)win * 96 + tid] = my_hd;
    for (int e = tid; e < XSTRIDE; e += 128) x[e] = xc[e];
    for (int l = tid; l < wm.L; l += 128) b.lam[wm.lm_off + l] = b.lamc[wm.lm_off + l];
  }
}

__global__ void k_init_state(BatchDev b, double radius0) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= b.W) return;
  SolverState &s = b.st[w];
  memset(&s, 0, sizeof(SolverState));
  s.radius = radius0;
  s.mu = 1e-8;
  s.need_lin = 1;
}

// =================================================================================================
// host-side launch sequence
// =================================================================================================
int vilo_launch_wave_solver(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s, int stage);   // kernels_wave.hip

int vilo_solve_launch(vilo_ctx *ctx, BatchDev &b, const vilo_solve_opts *o) {
  const double sq = ctx->cfg.focal_length / 1.5, ha = ctx->cfg.huber_delta, gn = ctx->cfg.g_norm;
  hipStream_t s = ctx->stream;
  static bool attr_set = false;
  const size_t lds_bytes = (size_t)LDS_TOTAL * sizeof(double);
  if (!attr_set) {
    VILO_HIP(hipFuncSetAttribute((const void *)k_build_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  SolveParams sp;
  sp.min_lm_diagonal = o->min_lm_diagonal; sp.max_lm_diagonal = o->max_lm_diagonal;
  sp.min_radius = o->min_trust_region_radius; sp.gradient_tolerance = o->gradient_tolerance;
  sp.jacobi_scaling = o->jacobi_scaling; sp.fixed_iterations = o->fixed_iterations;
  AcceptParams ap;
  ap.min_relative_decrease = o->min_relative_decrease; ap.function_tolerance = o->function_tolerance;
  ap.parameter_tolerance = o->parameter_tolerance; ap.max_num_iterations = o->max_num_iterations;
  ap.fixed_iterations = o->fixed_iterations; ap.init_mode = 1; ap.pad = 0;
  const int W = b.W;
  // VILO_SOLVER=fourwave: the round-1 one-workgroup-per-window solver (k_build_solve) instead of k_assemble_pose + k_solve_wave
  const char *solver_env = getenv("VILO_SOLVER");
  const bool wave_solver = !(solver_env && strcmp(solver_env, "fourwave") == 0);
  int pidx = 0;
  ctx->pev_kind.clear();
  auto P0 = [&](int kind) {
    if (!ctx->profile) return;
    while ((int)ctx->pev.size() < 2 * (pidx + 1)) { hipEvent_t e; (void)hipEventCreate(&e); ctx->pev.push_back(e); }
    ctx->pev_kind.push_back(kind);
    (void)hipEventRecord(ctx->pev[2 * pidx], s);
  };
  auto P1 = [&]() {
    if (!ctx->profile) return;
    (void)hipEventRecord(ctx->pev[2 * pidx + 1], s);
    ++pidx;
  };
  P0(6);
  hipLaunchKernelGGL(k_init_state, dim3((W + 127) / 128), dim3(128), 0, s, b, o->initial_trust_region_radius);
  P1();
  // IterationZero: cost at the initial point
  VILO_HIP(hipMemcpyAsync(b.xc, b.x, sizeof(double) * (size_t)W * XSTRIDE, hipMemcpyDeviceToDevice, s));
  P0(3);
  if (b.n_waves > 0) hipLaunchKernelGGL(k_visual_cost, dim3(b.n_waves, VILO_MAX_FRAMES), dim3(64), 0, s, b, sq, ha, 1);
  P1();
  P0(4);
  hipLaunchKernelGGL(k_imu_cost, dim3((W * 10 + 63) / 64), dim3(64), 0, s, b, gn, 1);
  P1();
  P0(5);
  hipLaunchKernelGGL(k_accept, dim3(W), dim3(128), 0, s, b, ap);
  P1();
  ap.init_mode = 0;
  for (int it = 0; it < o->max_num_iterations; ++it) {
    P0(0);
    launch_visual_linearize(b, sq, ha, s);
    P1();
    P0(7);
    hipLaunchKernelGGL(k_imu_raw, dim3((W * 10 + 63) / 64), dim3(64), 0, s, b, gn);
    P1();
    P0(1);
    hipLaunchKernelGGL(k_imu_whiten, dim3(W * 10), dim3(64), 0, s, b);
    P1();
    if (wave_solver) {
      P0(8);
      if (vilo_launch_wave_solver(ctx, b, sp, s, 0) != VILO_OK) return VILO_ERR_HIP;
      P1();
      P0(9);
      if (vilo_launch_wave_solver(ctx, b, sp, s, 1) != VILO_OK) return VILO_ERR_HIP;
      P1();
    } else {
      P0(2);
      hipLaunchKernelGGL(k_build_solve, dim3(W), dim3(SOLVE_THREADS), lds_bytes, s, b, sp);
      P1();
    }
    P0(3);
    if (b.n_waves > 0) hipLaunchKernelGGL(k_visual_cost, dim3(b.n_waves, VILO_MAX_FRAMES), dim3(64), 0, s, b, sq, ha, 0);
    P1();
    P0(4);
    hipLaunchKernelGGL(k_imu_cost, dim3((W * 10 + 63) / 64), dim3(64), 0, s, b, gn, 0);
    P1();
    P0(5);
    hipLaunchKernelGGL(k_accept, dim3(W), dim3(128), 0, s, b, ap);
    P1();
  }
  VILO_HIP(hipGetLastError());
  return VILO_OK;
}

// preMarginalize (marginalization_factor.cpp:119-138): evaluate the factors once at the current state.
int vilo_marg_linearize(vilo_ctx *ctx, BatchDev &b) {
  const double sq = ctx->cfg.focal_length / 1.5, ha = ctx->cfg.huber_delta, gn = ctx->cfg.g_norm;
  hipLaunchKernelGGL(k_init_state, dim3((b.W + 127) / 128), dim3(128), 0, ctx->stream, b, 1e4);
  launch_visual_linearize(b, sq, ha, ctx->stream);
  hipLaunchKernelGGL(k_imu_raw, dim3((b.W * 10 + 63) / 64), dim3(64), 0, ctx->stream, b, gn);
  hipLaunchKernelGGL(k_imu_whiten, dim3(b.W * 10), dim3(64), 0, ctx->stream, b);
  VILO_HIP(hipGetLastError());
  return VILO_OK;
}
