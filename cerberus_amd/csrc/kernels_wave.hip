// Single-wave solver for gfx950: what ceres::Solve does per linearisation for Estimator::optimization()
// (estimator.cpp:1221-1236; DENSE_SCHUR + traditional DOGLEG, Ceres 1.14 semantics), one WAVE per window.
//
//   k_assemble        (23-column Gram slots: batches in which a window estimates td; compact slots — every configuration of the reference —
//                     are assembled by kernels_asm_full.hip / kernels_asm_small.hip)
//                     one 256-thread workgroup per window (four per CU): the camera-side normal equations of the window in the layouts
//                     the solver streams. Pose / extrinsic / td system (80 x 80): owner-computes scatter (no atomics) of the
//                     per-(start frame, t) Gram slots of k_visual_linearize, the pose blocks of the IMU factor Grams and the prior
//                     image into an LDS image of 15 lower 16 x 16 tiles, written out in FP64-MFMA accumulator order (one coalesced
//                     load per register in the solver). Speed / leg-bias part: diagonal blocks A_kk, off-diagonal blocks (transposed),
//                     IMU coupling blocks with poses k-1 .. k+1, prior coupling rows, the diagonal and the gradient — constant blocks
//                     and absent frames already masked, so the solver has no index arithmetic in its loops.
//   k_solve_wave      one 64-lane workgroup per window, 17 KB of LDS, all 512 registers of its SIMD: four windows per CU, one per SIMD,
//                     no workgroup barriers and no idle waves. The 80 x 80 pose system lives in accumulator registers from its load to
//                     the backward solve.
//                     Jacobi scaling / dogleg diagonal / q = |J D^-2 g|^2  ->  block-tridiagonal Cholesky chain of the speed / leg-bias
//                     part with its coupling rows T(k) and C -= T^T T on the FP64 matrix cores  ->  landmark Schur complement on the
//                     matrix cores straight from global memory  ->  blocked 80 x 80 Cholesky with the forward solve riding along  ->
//                     backward solve from the registers  ->  bias and landmark back-substitution  ->  dogleg step and candidate state.
#include <type_traits>
#include "solve_common.hpp"
#include "chain_common.hpp"
#include "lin_common.hpp"

using namespace vilo;

// =================================================================================================
// k_assemble
// =================================================================================================
// position of pose-system entry (row, col) (tile row >= tile column) in the accumulator-order image: tile, register row >> 2, lane
// The LDS copy the scatter works on is the packed lower triangle of the 80 x 80 system (row r starts at r (r + 1) / 2: three integer
// operations per target instead of the tile arithmetic, 26 KB instead of 30, no mirror writes inside diagonal tiles); it is unpacked into
// the solver's tile image (15 lower 16 x 16 tiles in accumulator order = each tile row-major, diagonal tiles with both triangles) on the
// way out.
// (CL_N, cl_pos: lin_common.hpp — the small-batch assembly of kernels_asm_small.hip uses the same packed image)

#define ASM_THREADS 256

__device__ __forceinline__ void assemble_body(BatchDev &b, int jacobi_scaling, double min_lm_diagonal, double max_lm_diagonal) {
  __shared__ double Cl[CL_N];
  __shared__ double gst[2 * 780];   // ring of two IMU factor Grams
  __shared__ double gl[CD_N], hd[CD_N], vS[CD_N], red[12];
  __shared__ unsigned chunk_tab[64];
  __shared__ short inv_pmap[CD_N];
  __shared__ unsigned char act[CD_N];   // cd_active per camera dimension (the predicate has two integer divisions: looked up, not recomputed per entry)
  const int win = blockIdx.x, tid = threadIdx.x;
  const SolverState &st = b.st[win];
  if (st.done || !st.need_lin) return;
  const WinMeta wm = b.win[win];
  const int F = wm.n_frames, cmask = wm.const_mask, kb = wm.pad, pn = wm.prior_n;
  const double *gs = b.gram + (size_t)wm.gram_off * VILO_GRAM;
  const double *igram = b.imu_gram + (size_t)win * 10 * 780;
  const double *pd = b.prior_dense + (size_t)win * PD_N;
  double *bimg = b.Bimg + (size_t)win * BI_N;

  // ---- the pose system starts from the prior's pre-assembled image (zeros without a prior) ----
  PCLK(if (tid == 0) b.st[win].phase_clk[36] = clock64());
  {
    // (all 13 loads of a thread in flight before the first LDS store)
    double pv[13];
#pragma unroll
    for (int u = 0; u < 13; ++u) {
      const int e = min(tid + ASM_THREADS * u, CL_N - 1);
      int row = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      while (((row + 1) * (row + 2)) / 2 <= e) ++row;
      while ((row * (row + 1)) / 2 > e) --row;
      pv[u] = pd[PD_C + row * PD_CLD + (e - (row * (row + 1)) / 2)];
    }
#pragma unroll
    for (int u = 0; u < 13; ++u) {
      const int e = tid + ASM_THREADS * u;
      if (e < CL_N) Cl[e] = pv[u];
    }
  }
  for (int e = tid; e < CD_N; e += ASM_THREADS) { inv_pmap[e] = -1; act[e] = cd_active(e, F, cmask) ? 1 : 0; }
  if (tid < min(wm.n_chunks, 64)) {
    const ChunkMeta cm = b.chunk[wm.chunk_off + tid];
    chunk_tab[tid] = (unsigned)cm.s | ((unsigned)cm.kmax << 8) | ((unsigned)(cm.gram_off - wm.gram_off) << 16);
  }
  __syncthreads();
  if (tid < pn) inv_pmap[b.prior_map[(size_t)win * 96 + tid]] = (short)tid;
  __syncthreads();
  // gradient starts from the prior's b0 + H dx (H dx at the current point was formed by k_accept when it evaluated this point's cost)
  for (int e = tid; e < CD_N; e += ASM_THREADS) {
    const int pi = inv_pmap[e];
    gl[e] = (pn > 0 && pi >= 0) ? b.prior_b0[(size_t)win * 96 + pi] + b.prior_hd[(size_t)win * 96 + pi] : 0.0;
  }
  __syncthreads();
  PCLK(if (tid == 0) b.st[win].phase_clk[37] = clock64());

  // ---- plain (non-atomic) read-modify-write scatter: every target has exactly one owner thread. Two packed Gram entries can hit the
  //      same target only if they are "twins" (the same local pair taken once in the pose_s block and once in the pose_j block; for IMU
  //      factors once in the frame-i half and once in the frame-j half of the previous factor), so a thread owns an entry and its twin ----
  auto rmw = [&](int hi, int lo, double v) { lds_add(&Cl[cl_pos(hi, lo)], v); };   // hi >= lo
  {
    // visual Gram slots: 246 owner groups, one per thread
    // V1 pose_s x pose_s (21, twin pose_j x pose_j), V2 pose_s x pose_j (36), V3 pose_s x rest (84, twin pose_j x rest),
    // V4 rest x rest (105); rest = ex0 (6) ex1 (6) td r = local columns 12 .. 25
    int a = 0, bc = 0, cls = 0;
    if (tid < 21) { cls = 1; int rem = tid; while (rem >= 6 - a) { rem -= 6 - a; ++a; } bc = a + rem; }
    else if (tid < 57) { cls = 2; a = (tid - 21) / 6; bc = 6 + (tid - 21) % 6; }
    else if (tid < 141) { cls = 3; a = (tid - 57) / 14; bc = 12 + (tid - 57) % 14; }
    else if (tid < 246) { cls = 4; int rem = tid - 141; while (rem >= 14 - a) { rem -= 14 - a; ++a; } bc = 12 + a + rem; a += 12; }
    const bool isg = (bc == 25), dead = (cls == 0) || (a == 25);
    // (a, bc index the 26-column view [pose_s 6 | pose_j 6 | ex0 6 | ex1 6 | td | r] of a slot; the packed slot has 23 columns and a sign)
    double sg1, sg2 = 1.0;
    const int e1 = gram26_index(min(a, 25), bc, sg1);
    const int e2 = (cls == 1) ? gram26_index(a + 6, bc + 6, sg2) : (cls == 3 ? gram26_index(a + 6, bc, sg2) : e1);
    auto restcd = [](int c) { return c < 18 ? CD_EX0 + c - 12 : (c < 24 ? CD_EX1 + c - 18 : CD_TD); };
    const int rb = (bc >= 12 && bc < 25) ? restcd(bc) : 0, ra_ = (a >= 12 && a < 25) ? restcd(a) : 0;
    // Walk the window's chunks (fixed s, t = 0 .. kmax-1), two per trip = 44 loads in flight. Per chunk:
    //  stage 1: the entry's own target depends on s only (V1, V3, V4): sum over t in a register, one read-modify-write
    //  stage 2: the j-dependent target (twin of V1 / V3, the entry itself for V2)
    const int nch = min(wm.n_chunks, 64);
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
          v1[c2][t] = sg1 * gs[(size_t)(sl0 + tc) * VILO_GRAM + e1];
          v2[c2][t] = sg2 * gs[(size_t)(sl0 + tc) * VILO_GRAM + e2];
        }
      }
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        if (ch0 + c2 >= nch) continue;
        const int s_ = cs2[c2], km = km2[c2];
        if (cls != 2 && !dead) {
          double sum = 0.0;
#pragma unroll
          for (int t = 0; t < 11; ++t) sum += (t < km) ? v1[c2][t] : 0.0;
          if (isg) lds_add(&gl[cls == 4 ? ra_ : 6 * s_ + a], sum);
          else if (cls == 1) rmw(6 * s_ + bc, 6 * s_ + a, sum);
          else if (cls == 3) rmw(rb, 6 * s_ + a, sum);
          else rmw(rb, ra_, sum);
        }
        if (cls >= 1 && cls <= 3) {
#pragma unroll
          for (int t = 1; t < 11; ++t) {
            if (t < km) {
              const int j_ = s_ + t;
              const double val = (cls == 2) ? v1[c2][t] : v2[c2][t];
              if (cls == 3 && isg) lds_add(&gl[6 * j_ + a], val);
              else if (cls == 1) rmw(6 * j_ + bc, 6 * j_ + a, val);
              else if (cls == 2) rmw(6 * j_ + (bc - 6), 6 * s_ + a, val);
              else rmw(rb, 6 * j_ + a, val);
            }
          }
        }
      }
    }
  }
  __syncthreads();   // the IMU owners below are different threads
  PCLK(if (tid == 0) b.st[win].phase_clk[38] = clock64());
  // ---- IMU factors. Factor k (frames k, k + 1) contributes a packed 39 x 39 Gram [pose_i 6 | speed / leg-bias_i 13 | pose_j 6 | speed /
  //      leg-bias_j 13 | r] (780 entries, contiguous). Frame by frame through a ring of two Grams in LDS: the factor's 780 entries come in
  //      with coalesced loads (the next factor's are in flight in registers), every consumer gathers from LDS, and what goes to the solver
  //      (A_kk, A_k+1,k^T, the coupling rows) leaves as coalesced stores. Per frame k: Gi = factor k (frame k is its "i"), Gj = factor k - 1.
  //        [0, 169)    A_kk             = prior (frame kb) + Gi[6 + .][6 + .] + Gj[25 + .][25 + .]
  //        [169, 338)  A_{k+1,k}^T      = Gi[6 + j][25 + i]
  //        [338, 626)  coupling rows    with poses k - 1 (Gj), k (Gi + Gj), k + 1 (Gi); rows 13 .. 15 zero padding
  //        pose blocks of factor k      -> the pose image (63 owner threads), diagonal and gradient of the frame's 13 dimensions (13 threads)
  {
    double *ring = gst;
    // pose-block owners: I1 pose_i x pose_i (21, twin +19), I3 pose gradient (6, twin +19), I4 pose_i x pose_j (36): threads 128 .. 190 of the last trip
    int pa = 0, pbc = 0, pcls = 0;
    {
      const int q = tid - 128;
      if (q >= 0 && q < 21) { pcls = 1; int rem = q; while (rem >= 6 - pa) { rem -= 6 - pa; ++pa; } pbc = pa + rem; }
      else if (q >= 21 && q < 27) { pcls = 3; pa = q - 21; pbc = 38; }
      else if (q >= 27 && q < 63) { pcls = 4; pa = (q - 27) / 6; pbc = 19 + (q - 27) % 6; }
    }
    const int pe1 = tri39(pa, pbc), pe2 = (pcls == 4) ? pe1 : tri39(pa + 19, pcls == 3 ? 38 : pbc + 19);
    // what this thread's three entries per frame read and write does not depend on the frame: indices worked out once
    //   kind 0 A_kk, 1 A_{k+1,k}^T, 2 coupling row, 3 nothing;  si / sj: entry of Gi / Gj (-1: none);  d1 / d2: dimensions (within their frame's
    //   13) whose activity masks the entry (d2 < 0: only d1);  dst: offset inside the frame's block of the target array
    int e_kind[3], e_si[3], e_sj[3], e_d1[3], e_d2[3], e_dst[3], e_df[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + ASM_THREADS * u;
      e_kind[u] = 3; e_si[u] = -1; e_sj[u] = -1; e_d1[u] = 0; e_d2[u] = -1; e_dst[u] = 0; e_df[u] = 1;
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
        e_kind[u] = (i < 13) ? 2 : 4;   // (4: a zero-padding row of the coupling block)
        e_d1[u] = min(i, 12); e_dst[u] = is; e_df[u] = df;
        if (df == 1) { e_si[u] = tri39(c, 6 + min(i, 12)); e_sj[u] = tri39(19 + c, 25 + min(i, 12)); }
        else if (df == 2) e_si[u] = tri39(6 + min(i, 12), 19 + c);
        else e_sj[u] = tri39(c, 25 + min(i, 12));
      }
    }
    double gp[4];
    auto prefetch = [&](int k) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int e = tid + ASM_THREADS * u; gp[u] = (e < 780) ? igram[k * 780 + e] : 0.0; }
    };
    if (F > 1) prefetch(0);
    // the prior's diagonal block of frame kb (169 entries, threads 0 .. 168) comes in before the loop
    const double prior_ad = (kb >= 0 && tid < 169) ? pd[PD_AD + 169 * kb + tid] : 0.0;
    const double prior_dg = (kb >= 0 && tid >= 200 && tid < 213) ? pd[PD_AD + 169 * kb + (tid - 200) * 14] : 0.0;
    for (int k = 0; k < F; ++k) {
      const bool has_i = k < F - 1, has_j = k >= 1;
      double *Gi = ring + 780 * (k & 1);
      const double *Gj = ring + 780 * ((k + 1) & 1);
      lds_barrier();   // (the readers of this ring slot — frame k - 1's "Gj" of two frames ago — are done)
      if (has_i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = tid + ASM_THREADS * u; if (e < 780) Gi[e] = gp[u]; }
      }
      lds_barrier();
      if (k + 1 < F - 1) prefetch(k + 1);
      const unsigned char *actk = act + CD_B0 + 13 * k;
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
        } else if (kind == 1) {
          if (k < 10) {
            double v = 0.0;
            if (has_i && actk[13 + e_d1[u]] && actk[e_d2[u]]) v = Gi[e_si[u]];
            bimg[BI_AOT + 169 * k + e_dst[u]] = v;
          }
        } else if (kind == 2 || kind == 4) {
          const int f = k - 1 + e_df[u];
          double v = 0.0;
          if (kind == 2 && f >= 0 && f < F && actk[e_d1[u]]) {
            if (has_i && e_si[u] >= 0) v += Gi[e_si[u]];
            if (has_j && e_sj[u] >= 0) v += Gj[e_sj[u]];
          }
          bimg[BI_BS + 288 * k + e_dst[u]] = v;
        }
      }
      if (pcls && has_i) {
        const double vm = Gi[pe1], vt = Gi[pe2];
        if (pcls == 1) { rmw(6 * k + pbc, 6 * k + pa, vm); rmw(6 * (k + 1) + pbc, 6 * (k + 1) + pa, vt); }
        else if (pcls == 3) { lds_add(&gl[6 * k + pa], vm); lds_add(&gl[6 * (k + 1) + pa], vt); }
        else rmw(6 * (k + 1) + (pbc - 19), 6 * k + pa, vm);
      }
      if (tid >= 200 && tid < 213) {
        // diagonal and gradient of the frame's speed / leg-bias dimensions (the prior's share of the gradient is there already)
        const int i = tid - 200, cd = CD_B0 + 13 * k + i;
        double h = 1.0, g = gl[cd];
        if (act[cd]) {
          h = (k == kb) ? prior_dg : 0.0;
          if (has_i) { h += Gi[tri39(6 + i, 6 + i)]; g += Gi[tri39(6 + i, 38)]; }
          if (has_j) { h += Gj[tri39(25 + i, 25 + i)]; g += Gj[tri39(25 + i, 38)]; }
        }
        hd[cd] = h;
        gl[cd] = g;
      }
    }
  }
  __syncthreads();
  PCLK(if (tid == 0) b.st[win].phase_clk[39] = clock64());
  // ---- diagonal of the pose part; gradient of all 224 camera dimensions (inactive ones zero) ----
  // (rows / columns of inactive dimensions become identity rows when the image is written out below)
  for (int cd = tid; cd < CD_N; cd += ASM_THREADS) {
    double g = gl[cd];
    if (cd < CD_B0) hd[cd] = act[cd] ? Cl[cl_pos(cd, cd)] : 1.0;
    else if (cd >= CD_B0 + 13 * F) hd[cd] = 1.0;   // (frames beyond the window, padding: the loop above did not visit them)
    if (!act[cd]) g = 0.0;
    gl[cd] = g;
    b.cam_gin[(size_t)win * CD_N + cd] = g;
  }
  __syncthreads();
  PCLK(if (tid == 0) b.st[win].phase_clk[40] = clock64());
  // ---- Jacobi scaling 1 / (1 + sqrt(H_ii)) frozen at the first linearisation, dogleg diagonal clamp(diag, 1e-6, 1e32) in the scaled
  //      space, v = D^-2 g (Ceres 1.14 TrustRegionMinimizer / DoglegStrategy) ----
  double part_gn = 0.0, part_gmax = 0.0, part_q = 0.0;
  for (int cd = tid; cd < CD_N; cd += ASM_THREADS) {
    double d = 1.0, ve = 0.0;
    const double ge = gl[cd];
    if (act[cd]) {
      double sc;
      double *cs = b.cam_scale + (size_t)win * CD_N + cd;
      if (!st.scale_ready) { sc = jacobi_scaling ? 1.0 / (1.0 + sqrt(hd[cd])) : 1.0; *cs = sc; }
      else sc = *cs;
      const double d2 = fmin(fmax(sc * sc * hd[cd], min_lm_diagonal), max_lm_diagonal);
      d = d2 / (sc * sc);
      ve = ge / d;
    }
    vS[cd] = ve;
    bimg[BI_DH2 + cd] = d;
    bimg[BI_V + cd] = ve;
    part_gn += ge * ve;
    part_gmax = fmax(part_gmax, fabs(ge));
  }
  __syncthreads();
  PCLK(if (tid == 0) b.st[win].phase_clk[41] = clock64());
  // ---- pose system out in accumulator order; q = v^T H v of the camera-side rows is summed while the blocks pass through registers ----
  for (int idx = tid; idx < CIMG_N; idx += ASM_THREADS) {
    const int t = idx >> 8, rr = (idx >> 4) & 15, cc = idx & 15;
    const int row = 16 * c_tI[t] + rr, col = 16 * c_tJ[t] + cc;
    const double v = (act[row] && act[col]) ? Cl[cl_pos(max(row, col), min(row, col))] : (row == col ? 1.0 : 0.0);
    b.Cimg[(size_t)win * CIMG_N + idx] = v;
    part_q += ((c_tI[t] == c_tJ[t]) ? 1.0 : 2.0) * vS[row] * v * vS[col];   // (a diagonal tile holds both triangles)
  }
  PCLK(if (tid == 0) b.st[win].phase_clk[42] = clock64());
  // ---- prior rows of the frame whose speed / leg-bias block the prior touches (rows 13 .. 15: zero padding) ----
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int e = tid + ASM_THREADS * u;
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
    for (int e = tid + 169 * F; e < 11 * 169; e += ASM_THREADS) { const int ij = e % 169; bimg[BI_AD + e] = (ij / 13 == ij % 13) ? 1.0 : 0.0; }
    for (int e = tid + 169 * max(F - 1, 0); e < 10 * 169; e += ASM_THREADS) bimg[BI_AOT + e] = 0.0;
    for (int e = tid + 288 * F; e < 11 * 288; e += ASM_THREADS) bimg[BI_BS + e] = 0.0;
  }
  PCLK(if (tid == 0) b.st[win].phase_clk[43] = clock64());
  // ---- q = v^T H v of the speed / leg-bias rows: the blocks just written come back (coalesced, L2-resident; all loads of a kind in flight) ----
  __threadfence_block();   // (this workgroup's stores above must be visible to its loads below)
  __syncthreads();
  {
    double val[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) val[u] = bimg[BI_AD + min(tid + ASM_THREADS * u, 11 * 169 - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + ASM_THREADS * u;
      if (e < 11 * 169) {
        const int k = e / 169, ij = e - 169 * k, i = ij / 13, j = ij - 13 * i;
        part_q += vS[CD_B0 + 13 * k + i] * val[u] * vS[CD_B0 + 13 * k + j];
      }
    }
  }
  {
    double val[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) val[u] = bimg[BI_AOT + min(tid + ASM_THREADS * u, 10 * 169 - 1)];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int e = tid + ASM_THREADS * u;
      if (e < 10 * 169) {
        const int k = e / 169, ji = e - 169 * k, j = ji / 13, i = ji - 13 * j;
        part_q += 2.0 * vS[CD_B0 + 13 * (k + 1) + i] * val[u] * vS[CD_B0 + 13 * k + j];
      }
    }
  }
  {
    double val[13];
#pragma unroll
    for (int u = 0; u < 13; ++u) val[u] = bimg[BI_BS + min(tid + ASM_THREADS * u, 11 * 288 - 1)];
#pragma unroll
    for (int u = 0; u < 13; ++u) {
      const int e = tid + ASM_THREADS * u;
      if (e < 11 * 288) {
        const int k = e / 288, is = e - 288 * k, i = is / 18, sx = is - 18 * i, f = k - 1 + sx / 6, c = sx % 6;
        if (val[u] != 0.0) part_q += 2.0 * vS[CD_B0 + 13 * k + min(i, 12)] * val[u] * vS[min(max(6 * f + c, 0), 79)];
      }
    }
  }
  PCLK(if (tid == 0) b.st[win].phase_clk[44] = clock64());
  // camera-side sums of |D^-1 g|^2, max |g| and q (the landmarks add theirs in the solver): waves in fixed order
  part_q = wave_sum(part_q); part_gn = wave_sum(part_gn); part_gmax = wave_max(part_gmax);
  if ((tid & 63) == 0) { red[tid >> 6] = part_q; red[4 + (tid >> 6)] = part_gn; red[8 + (tid >> 6)] = part_gmax; }
  __syncthreads();
  PCLK(if (tid == 0) b.st[win].phase_clk[45] = clock64());
  if (tid == 0) {
    bimg[BI_SCAL + 0] = ((red[0] + red[1]) + red[2]) + red[3];
    bimg[BI_SCAL + 1] = ((red[4] + red[5]) + red[6]) + red[7];
    bimg[BI_SCAL + 2] = fmax(fmax(red[8], red[9]), fmax(red[10], red[11]));
  }
}
__global__ void __launch_bounds__(ASM_THREADS) k_assemble(BatchDev b, int jacobi_scaling, double min_lm_diagonal, double max_lm_diagonal) {
  assemble_body(b, jacobi_scaling, min_lm_diagonal, max_lm_diagonal);
}

// =================================================================================================
// k_solve_wave
// =================================================================================================
// LDS map (doubles; 2176 = 17408 B per workgroup: eight workgroups per CU = two per SIMD, the register file's limit at 256 VGPRs — the
// latency-bound phases of one window (readlane chains of the 13 x 13 / 16 x 16 factorisations, substitutions, memory round trips) run
// under the matrix-core phases of the other). The pose system and its Cholesky factor live in REGISTERS from the tile load to the
// backward solve; LDS only stages what changes layout between accumulator order and operand order.
#define WS_C 0          // 1024: scratch region with three lives
#define WS_G 1024
#define WS_DH2 1104
#define WS_Y 1184
#define WS_V 1264
#define WS_SCR 1344
#define WS_TOTAL 2176
// (1) chain + Schur pass: vectors of the speed / leg-bias part
#define WC_DB 144       // [144] dhat^2
#define WC_GB 288       // [144] gradient
// (2) Cholesky: the panel tiles L_Ij (I > j) of the current block column, slot I - j - 1, 256 each, element (r, c) of a tile at
//     16 r + ((c + r) & 15): conflict-free for accumulator-order stores (row lk + 4 reg, column lr) and operand-order reads (row lr, column 4 kk + lk)
// (3) back-substitution of the speed / leg-bias part: M_k and T_A of the frame in flight
#define WB_M 0          // [169]
#define WB_TA 176       // [169]
// scratch during the chain
#define WX_LM 0         // 13 x 13: M_k = L_k^-1
#define WX_TA0 176
#define WX_TA1 352
#define WX_SN 528       // 13 x 13: S_{k-1} = A_{k-1,k-1} - T_A(k)^T T_A(k)
// scratch during the Cholesky
#define WX_D16 0        // 16 x 17
#define WX_LI16 272     // 16 x 17 + 16
#define WX_P16 560      // 16 x 17
// back-substitution of the speed / leg-bias part, step
#define WX_U 0          // [144]
#define WX_YB 144       // [144]
#define WX_DEL 288      // [224] step of the camera dimensions

extern "C" size_t vilo_solve_wave_lds_bytes() { return (size_t)WS_TOTAL * sizeof(double); }

// MID = false: the complete solve of a linearisation (k_solve_wave; redo_only: only for the windows the three-stage form flagged).
// MID = true: middle stage of the three-stage form (k_solve_mid, kernels_split.hip): the chain's results come from k_chain, the kernel
// stops after the backward solve (y_P to cam_y; k_backsub goes on). A window whose chain or Cholesky-80 failed is flagged for the complete
// path (SolverState::pad[1] = 4), which finds the same failure and takes DoglegStrategy::ComputeGaussNewtonStep's retry loop from there.
template <bool MID>
__device__ __forceinline__ void solve_wave_body(BatchDev &b, const SolveParams &sp, int redo_only, double *lds) {
  const int win = blockIdx.x;
  SolverState &st = b.st[win];
  if (st.done) return;
  if (MID && !st.need_lin) return;     // (k_backsub takes the dogleg step of a reused linearisation)
  if (!MID && redo_only && st.pad[1] != 4) return;
  const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
  const WinMeta wm = b.win[win];
  const int F = wm.n_frames, L = wm.L, kb = wm.pad, cmask = wm.const_mask;
  double *g = lds + WS_G, *dh2 = lds + WS_DH2, *y = lds + WS_Y, *v = lds + WS_V, *scr = lds + WS_SCR;
  // gradient, dogleg diagonal and Gauss-Newton step of the 143 speed / leg-bias dimensions: dimension lane + 64 m in register m
  double gBr[3], dBr[3], yBr[3];

  if (st.need_lin) {
    PCLK(if (lane == 0) st.phase_clk[0] = clock64());
    const double *Cimg = b.Cimg + (size_t)win * CIMG_N;
    const double *gin = b.cam_gin + (size_t)win * CD_N;
    const double *bimg = b.Bimg + (size_t)win * BI_N;
    const double *wl = b.lm_w + 80 * (size_t)wm.lm_off;
    double *lm_E = b.lm_E + wm.lm_off, *lm_g = b.lm_gbuf[st.cur] + wm.lm_off, *lm_dh2 = b.lm_dh2 + wm.lm_off, *lm_scale = b.lm_scale + wm.lm_off,
           *lm_einv = b.lm_einv + wm.lm_off, *lm_y = b.lm_y + wm.lm_off;
    double *Mg = b.Lk + (size_t)win * 11 * 169, *TAg = b.TAg + (size_t)win * 11 * 169;
    const bool first_scale = !st.scale_ready;
    double mu = st.mu;

    // ---- gradient, dogleg diagonal and v = D^-2 g of the camera dimensions come scaled from k_assemble (with their share of q, |D^-1 g|^2
    //      and max |g|); the landmarks are scaled here ----
    double part_gn = 0.0, part_gmax = 0.0, part_q = 0.0;
    for (int cd = lane; cd < 80; cd += 64) { g[cd] = gin[cd]; dh2[cd] = bimg[BI_DH2 + cd]; v[cd] = bimg[BI_V + cd]; }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int e = min(lane + 64 * m, 143);
      gBr[m] = gin[CD_B0 + e]; dBr[m] = bimg[BI_DH2 + CD_B0 + e]; yBr[m] = 0.0;
      if (lane + 64 * m < 144) { lds[WC_DB + e] = dBr[m]; lds[WC_GB + e] = gBr[m]; }
    }
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
    const double gnorm2 = bimg[BI_SCAL + 1] + wave_sum(part_gn), gmax = fmax(bimg[BI_SCAL + 2], wave_max(part_gmax));
    if (!sp.fixed_iterations && gmax <= sp.gradient_tolerance) {
      if (lane == 0) { st.gmax = gmax; st.done = 1; st.termination = 1; st.step_valid = 0; }
      return;
    }
    lds_fence();

    bool solved = false, have_q = false;
    double qq = 0.0, gnnorm2 = 0.0, gy = 0.0;
    constexpr bool pre = MID;
    if (MID) {
      const int chain_ok = st.pad[1] == 1;
      if (!__builtin_amdgcn_readfirstlane(chain_ok)) {   // a pivot of the chain was not positive at this mu
        if (lane == 0) st.pad[1] = 4;
        return;
      }
    }
    const int lane_outer = lane;
    while (!solved) {
      // The body almost never repeats (only when a factorisation fails and mu grows). The lane index is made opaque per trip so that the
      // compiler does not hoist every lane-derived address of the body out of the "loop" and spill it across the whole kernel.
      int lane = lane_outer;
      asm volatile("" : "+v"(lane));
      const int lr = lane & 15, lk = lane >> 4;
      // per-lane constants of the coupling-block gathers: column 16 X + lr of the pose system is column oX of pose fX
      int fX[5], oX[5];
#pragma unroll
      for (int X = 0; X < 5; ++X) { const int col = 16 * X + lr; fX[X] = col < 66 ? col / 6 : 99; oX[X] = col < 66 ? col - 6 * fX[X] : 0; }
      PCLK(if (lane == 0) st.phase_clk[1] = clock64());
      for (int l = lane; l < L; l += 64) lm_einv[l] = 1.0 / (lm_E[l] + mu * lm_dh2[l]);
      PCLK(if (lane == 0) st.phase_clk[2] = clock64());

      // ---- pose system: 15 lower tiles in accumulator order, one coalesced load per register ----
      mfma_d4 acc[15];
#pragma unroll
      for (int t = 0; t < 15; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = Cimg[(t * 4 + r) * 64 + lane];
      int fail = 0;
      if constexpr (MID) {
        // the chain's part comes from k_chain (chain_common.hpp) through global memory: T_B(k) in operand order, the reduced right-hand side
        const double *Tg = b.Tk + (size_t)win * TK_N;
        // C -= T_B(k)^T T_B(k) for frames F-1 .. 0, operands as the chain left them (L2-resident; the next frame's in flight)
        for (int cd = lane; cd < 80; cd += 64) v[cd] = Tg[TK_V + cd];
        double tb[2][5][4];
        // (one branch-free form per x_lo: conditions between the MFMAs of a frame stall the stream)
        auto ldT_x = [&](int k, auto selc, auto xc) {
          constexpr int sel = decltype(selc)::value, XLO = decltype(xc)::value;
#pragma unroll
          for (int X = XLO; X < 5; ++X)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) tb[sel][X][kk] = (kk < 3 || lk == 0) ? Tg[1280 * k + (X * 4 + kk) * 64 + lane] : 0.0;   // (rows 13 .. 15: padding)
        };
        auto upd_x = [&](auto selc, auto xc) {
          constexpr int sel = decltype(selc)::value, XLO = decltype(xc)::value;
#pragma unroll
          for (int t = 0; t < 15; ++t) {
            if (c_tJ[t] >= XLO) {
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-tb[sel][c_tI[t]][kk], tb[sel][c_tJ[t]][kk], acc[t], 0, 0, 0);
            }
          }
        };
        auto ldT = [&](int k, auto selc) {
          switch (chain_x_lo(k, kb)) {
            case 0: ldT_x(k, selc, std::integral_constant<int, 0>{}); break;
            case 1: ldT_x(k, selc, std::integral_constant<int, 1>{}); break;
            case 2: ldT_x(k, selc, std::integral_constant<int, 2>{}); break;
            default: ldT_x(k, selc, std::integral_constant<int, 3>{}); break;
          }
        };
        auto upd = [&](int k, auto selc) {
          switch (chain_x_lo(k, kb)) {
            case 0: upd_x(selc, std::integral_constant<int, 0>{}); break;
            case 1: upd_x(selc, std::integral_constant<int, 1>{}); break;
            case 2: upd_x(selc, std::integral_constant<int, 2>{}); break;
            default: upd_x(selc, std::integral_constant<int, 3>{}); break;
          }
        };
        std::integral_constant<int, 0> s0;
        std::integral_constant<int, 1> s1;
        ldT(F - 1, s0);
        for (int k = F - 1; k >= 0; k -= 2) {
          if (k >= 1) ldT(k - 1, s1);
          upd(k, s0);
          if (k >= 1) {
            if (k >= 2) ldT(k - 2, s0);
            upd(k - 1, s1);
          }
        }
      } else {
      // ---- block-tridiagonal Cholesky chain of the speed / leg-bias part (13 x 13 blocks, frames F-1 .. 0):
      //        S_k = A_kk + mu D_k - T_A(k+1)^T T_A(k+1),  L_k = chol(S_k),  M_k = L_k^-1,  T_A(k) = M_k A_{k,k-1},
      //        V = [B_k | g_k] - T_A(k+1)^T T(k+1),  T(k) = M_k V   (13 x 80: columns 0..78 coupling rows, column 79 = rhs),
      //        C -= T_B(k)^T T_B(k),  rhs_P -= T_B(k)^T t_g(k).
      //      The scalar part runs lane = row in the four 16-lane groups; T, V and the rank update are FP64-MFMA tiles whose accumulator
      //      layout (register r of lane (lr, lk) = row lk + 4 r, column lr) is the operand layout of the next product. ----
      {
        const int grp = lk, c = lr;
        const int row = c < 13 ? c : 0;
        double *LM = scr + WX_LM, *SN = scr + WX_SN;
        double *TAcur = scr + WX_TA0, *TAprev = scr + WX_TA1;
        const double *DB = lds + WC_DB, *GB = lds + WC_GB;
        mfma_d4 T[5];
        double yr[5];
#pragma unroll
        for (int X = 0; X < 5; ++X) { T[X] = mfma_d4{0.0, 0.0, 0.0, 0.0}; yr[X] = 0.0; }
        // this frame's blocks come from the assembled image one frame ahead of their use (registers nV / nrhs / nadn)
        auto load_blocks = [&](int k, mfma_d4 *Vn, double *rhsn, double *adnn) {
          // [B_k | g_k] in accumulator order: row lk + 4 r, column 16 X + lr (zero rows 13..15 in the image); column 79 carries the gradient
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
          for (int r = 0; r < 4; ++r) adnn[r] = (k > 0 && lr < 13 && lk + 4 * r < 13) ? bimg[BI_AD + max(k - 1, 0) * 169 + (lk + 4 * r) * 13 + lr] : 0.0;   // A_{k-1,k-1}, accumulator order
        };
        mfma_d4 nV[5];
        double nrhs[13], nadn[4];
        load_blocks(F - 1, nV, nrhs, nadn);
        for (int k = F - 1; k >= 0; --k) {
          const int x_lo = (k <= kb) ? 0 : max(0, (6 * (k - 1)) >> 4);   // T(k) is zero left of pose k - 1 (dense from the prior's frame down)
          mfma_d4 V[5];
          double a[13], l[13], rhs[13], adn[4];
#pragma unroll
          for (int X = 0; X < 5; ++X) V[X] = nV[X];
#pragma unroll
          for (int i = 0; i < 13; ++i) rhs[i] = nrhs[i];
#pragma unroll
          for (int m = 0; m < 4; ++m) adn[m] = nadn[m];
          if (lr == 15) {
#pragma unroll
            for (int r = 0; r < 4; ++r) V[4][r] = (lk + 4 * r < 13) ? GB[13 * k + lk + 4 * r] : 0.0;
          }
          // S_k (lane = row): the top frame straight from A_kk, later frames from the update left by the previous step
          if (k == F - 1) {
#pragma unroll
            for (int j = 0; j < 13; ++j) a[j] = bimg[BI_AD + (k * 13 + row) * 13 + j];
          } else {
#pragma unroll
            for (int j = 0; j < 13; ++j) a[j] = SN[row * 13 + j];
          }
          PCLK(if (k == 5 && lane == 0) st.phase_clk[16] = clock64());
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
          PCLK(if (k == 5 && lane == 0) st.phase_clk[17] = clock64());
          // forward substitutions L x = rhs: T_A(k) columns (group 0), L^-1 columns (group 1); L broadcast from the owning lanes
          // (opaque copies: see chol16_tile)
#pragma unroll
          for (int j = 0; j < 13; ++j) asm volatile("" : "+v"(l[j]));
          double cl[13];
#pragma unroll
          for (int i = 0; i < 13; ++i) {
            double vv = (grp == 0) ? rhs[i] : ((i == c) ? 1.0 : 0.0);
#pragma unroll
            for (int q = 0; q < i; ++q) vv -= readlane_d(l[q], i) * cl[q];
            cl[i] = vv * readlane_d(myrinv, i);
            asm volatile("" : "+v"(cl[i]));
            __builtin_amdgcn_sched_barrier(0);   // (one row's v_readlane results at a time)
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
          PCLK(if (k == 5 && lane == 0) st.phase_clk[18] = clock64());
          // S_{k-1} = A_{k-1,k-1} - T_A(k)^T T_A(k): one 16 x 16 tile on the matrix cores (the operand serves as A and B)
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
          PCLK(if (k == 5 && lane == 0) st.phase_clk[19] = clock64());
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
          // the next frame's blocks: in flight behind the rank update below (V, rhs and adn of this frame are dead)
          if (k > 0) load_blocks(k - 1, nV, nrhs, nadn);
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
          for (int t = 0; t < 15; ++t) {
            const int I = c_tI[t], J = c_tJ[t];
            if (J >= x_lo) {
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) {
                const double opa = (I == 4) ? T4[kk] : T[I][kk], opb = (J == 4) ? T4[kk] : T[J][kk];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-opa, opb, acc[t], 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int X = 0; X < 5; ++X)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) yr[X] += ((X == 4) ? T4[kk] : T[X][kk]) * tg[kk];
          PCLK(if (k == 5 && lane == 0) st.phase_clk[20] = clock64());
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
      }
      PCLK(if (lane == 0) st.phase_clk[3] = clock64());

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
        // A landmark that starts in frame s couples with poses s .. only: rows 0 .. 6 s - 1 of its coupling column are structural zeros
        // (k_visual_linearize writes them as such), and the landmarks of a window are ordered by start frame. A trip of 16 landmarks
        // whose first one starts in frame s >= 3 has nothing in tile row 0 (rows 0 .. 15): neither loaded nor multiplied — 10 tiles
        // instead of 15. Two branch-free forms of the trip, chosen per trip (per-tile conditions inside one form stall the MFMA stream).
        int *skip_tab = (int *)(scr + WX_LM);   // (the chain's scratch is dead)
        const int ntrip = (L + 15) >> 4;
        if (L > 0) {
          const unsigned char *lms = b.lm_s + wm.lm_off;
          for (int tr = lane; tr < ntrip + 2; tr += 64) skip_tab[tr] = (6 * (int)lms[min(16 * tr, L - 1)] >= 16) ? 1 : 0;
          lds_fence();
        }
        double opb[2][4][5], eb[2][4], gb[2][4], db[2][4];
        auto ldtrip = [&](int kk0, int bsel) {
          const int skip0 = __builtin_amdgcn_readfirstlane(skip_tab[min(kk0 >> 2, ntrip)]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int l = 4 * (kk0 + u) + lk, lc = min(l, L - 1);
            eb[bsel][u] = lm_einv[lc]; gb[bsel][u] = lm_g[lc]; db[bsel][u] = lm_y[lc];
#pragma unroll
            for (int X = 1; X < 5; ++X) opb[bsel][u][X] = wl[(size_t)(16 * X + lr) * L + lc];
          }
          if (!skip0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) opb[bsel][u][0] = wl[(size_t)lr * L + min(4 * (kk0 + u) + lk, L - 1)];
          }
        };
        auto dotrip = [&](int kk0, int bsel, auto xl) {
          constexpr int XL = decltype(xl)::value;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int l = 4 * (kk0 + u) + lk;
            const double ei = (l < L) ? eb[bsel][u] : 0.0, ge = gb[bsel][u] * ei, vl = (l < L) ? db[bsel][u] : 0.0;
            double op[5];
#pragma unroll
            for (int X = 0; X < 5; ++X) op[X] = (X >= XL) ? opb[bsel][u][X] * actv[X] : 0.0;
#pragma unroll
            for (int t = 0; t < 15; ++t)
              if (c_tJ[t] >= XL) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-(op[c_tI[t]] * ei), op[c_tJ[t]], acc[t], 0, 0, 0);
#pragma unroll
            for (int X = XL; X < 5; ++X) { yacc[X] += op[X] * ge; qacc += op[X] * vv[X] * vl; }
          }
        };
        auto trip = [&](int kk0, int bsel) {
          const int skip0 = __builtin_amdgcn_readfirstlane(skip_tab[min(kk0 >> 2, ntrip)]);
          if (skip0) dotrip(kk0, bsel, std::integral_constant<int, 1>{});
          else dotrip(kk0, bsel, std::integral_constant<int, 0>{});
        };
        if (L > 0) {
          // the landmark vectors written above (lm_einv, lm_y) are read back through global memory by other lanes
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if constexpr (MID) {
            ldtrip(0, 0);
            for (int kk0 = 0; kk0 < nks; kk0 += 8) {
              if (kk0 + 4 < nks) ldtrip(kk0 + 4, 1);
              trip(kk0, 0);
              if (kk0 + 4 < nks) {
                if (kk0 + 8 < nks) ldtrip(kk0 + 8, 0);
                trip(kk0 + 4, 1);
              }
            }
          } else {
            // (unguarded: loads past the last landmark are clamped, their products masked; the guards cost this instantiation registers)
            ldtrip(0, 0);
            for (int kk0 = 0; kk0 < nks; kk0 += 8) {
              ldtrip(kk0 + 4, 1);
              trip(kk0, 0);
              ldtrip(kk0 + 8, 0);
              trip(kk0 + 4, 1);
            }
          }
        }
        if (!have_q) {
          part_q += 2.0 * qacc;
          qq = bimg[BI_SCAL + 0] + wave_sum(part_q);
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
      PCLK(if (lane == 0) st.phase_clk[4] = clock64());

      // ---- dense Cholesky of the 80 x 80 reduced pose system, blocked by 16: diagonal tile in registers + v_readlane (also its
      //      inverse), panel L_Ij = A_Ij L_jj^-T and trailing update A_IJ -= L_Ij L_Jj^T on the FP64 matrix cores. The reduced right-hand
      //      side rides along as a sixth block row (one row of a tile), which makes it y = L^-1 rhs by the end: the forward solve costs
      //      the same 60 MFMAs and the factor never has to be read in operand order again — it stays in the accumulator registers
      //      (L_Ij, and L_jj^-1 in the diagonal tiles), which IS the operand order of L^T for the backward solve ----
      {
        double *D16 = scr + WX_D16, *LI16 = scr + WX_LI16, *P16 = scr + WX_P16;
        double vrow[5];   // lane (lr, lk == 0): entry 16 J + lr of the right-hand-side row
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
            for (int r = 0; r < 4; ++r) lds[pswz(I - j - 1, lk + 4 * r, lr)] = nacc[r];
            lds_fence();   // (P16 is reused by the next panel)
          }
          // right-hand-side row: y_j^T = rhs_j^T L_jj^-T; its operand-order copy (row 0 of a tile) through the LDS vector y
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
          // trailing update: operand (row lr, columns 4 kk + lk) of every panel tile once
          double pa[5][4];
#pragma unroll
          for (int I = j + 1; I < 5; ++I)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) pa[I][kk] = lds[pswz(I - j - 1, lr, 4 * kk + lk)];
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
      if (MID && fail) {
        if (lane == 0) st.pad[1] = 4;
        return;
      }
      if (fail) {
        // DoglegStrategy::ComputeGaussNewtonStep: mu *= 10 and retry while mu < max_mu (1.0)
        mu *= 10.0;
        if (lane == 0) { st.mu = mu; st.pad[0]++; }   // (pad[0]: factorisation retries of this solve, read by the tests)
        if (!(mu < 1.0)) {
          if (lane == 0) { st.lin_fail = 1; st.step_valid = 0; st.gnorm2 = gnorm2; st.q = qq; st.gmax = gmax; st.scale_ready = 1; st.pad[1] = 3; }
          return;
        }
        // the vectors of the speed / leg-bias part shared the C region with the factor: put them back
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const int e = lane + 64 * m;
          if (e < 144) { lds[WC_DB + e] = dBr[m]; lds[WC_GB + e] = gBr[m]; }
        }
        lds_fence();
        continue;
      }
      PCLK(if (lane == 0) st.phase_clk[5] = clock64());

      // ---- L^T yP = y (the forward solve rode along with the factorisation) ----
      {
        // Blockwise on the FP64 matrix cores: x_j = L_jj^-T (y_j - sum_{i>j} L_ij^T x_i). A block vector lives in accumulator order,
        // replicated over the 16 columns (register r of lane (lr, lk) = entry lk + 4 r), which is the B-operand order of the next
        // product; the A operands are the factor's accumulator registers as they stand.
        mfma_d4 yb[5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) yb[j][r] = y[16 * j + lk + 4 * r];
#pragma unroll
        for (int j = 4; j >= 0; --j) {
          mfma_d4 accv = yb[j];
#pragma unroll
          for (int i = j + 1; i < 5; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) accv = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[tile_index(i, j)][kk], yb[i][kk], accv, 0, 0, 0);
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
      lds_fence();
      PCLK(if (lane == 0) st.phase_clk[6] = clock64());
      if constexpr (MID) {
        // hand-over to k_backsub: y_P and the scalars of this linearisation
        double *cam_y = b.cam_y + (size_t)win * CD_N;
        for (int cd = lane; cd < 80; cd += 64) cam_y[cd] = y[cd];
        if (lane == 0) { st.gnorm2 = gnorm2; st.q = qq; st.gmax = gmax; st.pad[1] = 2; }
        return;
      }

      // ---- back-substitution of the speed / leg-bias part: c = g_B - B yP, then the two block-bidiagonal sweeps
      //        u_k = M_k (c_k - T_A(k+1)^T u_{k+1})   k = F-1 .. 0,      y_k = M_k^T (u_k - T_A(k) y_{k-1})   k = 0 .. F-1 ----
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
        double *U = scr + WX_U, *YB = scr + WX_YB;
        PCLK(if (lane == 0) st.phase_clk[21] = clock64());
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
        PCLK(if (lane == 0) st.phase_clk[22] = clock64());
        // M_k / T_A(k) of the chain (written to global memory by this wave, L2-resident) come back one frame at a time: the next
        // frame's 2 x 169 values are in flight while this frame's are used out of LDS
        double *MB = lds + WB_M, *TB = lds + WB_TA;
        const double *Mg_ = b.Lk + (size_t)win_b * 11 * 169, *TAg_ = b.TAg + (size_t)win_b * 11 * 169;
        int pe[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) pe[m] = min(lane + 64 * m, 168);
        double pm[3], pt[3];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < 3; ++m) { pm[m] = Mg_[(F - 1) * 169 + pe[m]]; pt[m] = 0.0; }
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
        PCLK(if (lane == 0) st.phase_clk[23] = clock64());
        // backward sweep (pm holds M_0; T_A(0) does not exist)
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
      }
      PCLK(if (lane == 0) st.phase_clk[7] = clock64());
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
        part_gnn += dh2[cd] * y[cd] * y[cd];
        part_gy += g[cd] * y[cd];
      }
      gnnorm2 = wave_sum(part_gnn);
      gy = wave_sum(part_gy);
      if (!(isfinite(gnnorm2) && isfinite(gy))) {   // IsArrayValid(gauss_newton_step_) failed
        mu *= 10.0;
        if (lane == 0) { st.mu = mu; st.pad[0]++; }   // (pad[0]: factorisation retries of this solve, read by the tests)
        if (!(mu < 1.0)) {
          if (lane == 0) { st.lin_fail = 1; st.step_valid = 0; st.scale_ready = 1; st.pad[1] = 3; }
          return;
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const int e = lane + 64 * m;
          if (e < 144) { lds[WC_DB + e] = dBr[m]; lds[WC_GB + e] = gBr[m]; }
        }
        lds_fence();
        continue;
      }
      solved = true;
    }
    // keep the linearisation's vectors for the steps that reuse it after a rejected candidate
    int win_c = win;
    asm volatile("" : "+s"(win_c));
    double *cam_g = b.cam_g + (size_t)win_c * CD_N, *cam_dh2 = b.cam_dh2 + (size_t)win_c * CD_N, *cam_y = b.cam_y + (size_t)win_c * CD_N;
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
      st.pad[1] = 3;
      PCLK(st.phase_clk[8] = clock64());
    }
  } else {
    const double *cam_g = b.cam_g + (size_t)win * CD_N, *cam_dh2 = b.cam_dh2 + (size_t)win * CD_N, *cam_y = b.cam_y + (size_t)win * CD_N;
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
  PCLK(if (lane == 0) st.phase_clk[9] = clock64());
}

__global__ void __launch_bounds__(64) k_solve_wave(BatchDev b, SolveParams sp, int redo_only) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  solve_wave_body<false>(b, sp, redo_only, lds);
}
__global__ void __launch_bounds__(64) k_solve_mid(BatchDev b, SolveParams sp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  solve_wave_body<true>(b, sp, 0, lds);
}

// =================================================================================================
// launch
// =================================================================================================
int vilo_launch_mw8_solver(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s);   // kernels_mw8.hip
// Which solver (VILO_SOLVER_*: 0 the single wave, 3 the single wave in three stages, 4 eight waves per window): as many waves per window as
// the batch leaves SIMDs for — eight up to two rounds of one window per CU (512 on an MI355X; measured against the single wave with the
// two-kernel assembly: 320 windows + 5 %, 384 + 7 %, 512 + 4 %), the single wave beyond, in three stages once the batch fills the
// two-waves-per-SIMD stages too (VILO_MW8_MAX_WINDOWS / VILO_SPLIT_MIN_WINDOWS move the two thresholds). vilo_set_solver_form
// pins a form (the tests run every form against the oracle; a deployment that needs bitwise equal answers across batch sizes pins one too).
int vilo_solver_form(const vilo_ctx *ctx, const BatchDev &b) {
  const int forced = ctx->solver_form;   // (vilo_set_solver_form; VILO_SOLVER gives the default at vilo_create)
  static const int max_w8 = [] { const char *e = getenv("VILO_MW8_MAX_WINDOWS"); return e ? atoi(e) : 512; }();
  static const int min_w3 = [] { const char *e = getenv("VILO_SPLIT_MIN_WINDOWS"); return e ? atoi(e) : 1025; }();
  if (forced >= 0) return forced;
  return b.W <= max_w8 ? 4 : (b.W < min_w3 ? 0 : 3);
}
int vilo_launch_wave_solver(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s, int stage) {
  size_t lds_bytes = (size_t)WS_TOTAL * sizeof(double);
  if (const char *e = getenv("VILO_WAVE_LDS")) lds_bytes = (size_t)atol(e);   // occupancy experiments: more LDS per workgroup = fewer windows per CU
  if (stage == 0) {
    hipLaunchKernelGGL(k_assemble, dim3(b.W), dim3(ASM_THREADS), 0, s, b, sp.jacobi_scaling, sp.min_lm_diagonal, sp.max_lm_diagonal);
  } else if (vilo_solver_form(ctx, b) == 4) {
    return vilo_launch_mw8_solver(ctx, b, sp, s);
  } else {
    if (!ctx->wave_attr_set) {
      VILO_HIP(hipFuncSetAttribute((const void *)k_solve_wave, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      ctx->wave_attr_set = true;
    }
    if (stage == 2 && !ctx->mid_attr_set) {   // (k_solve_mid takes the same dynamic LDS, VILO_WAVE_LDS included)
      VILO_HIP(hipFuncSetAttribute((const void *)k_solve_mid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      ctx->mid_attr_set = true;
    }
    // stage 1: the complete single-wave solver; stages 2 .. 4: the rest of the three-stage form (k_chain is stage 5 in kernels_split.hip)
    if (stage == 1) hipLaunchKernelGGL(k_solve_wave, dim3(b.W), dim3(64), lds_bytes, s, b, sp, 0);
    else if (stage == 2) hipLaunchKernelGGL(k_solve_mid, dim3(b.W), dim3(64), lds_bytes, s, b, sp);
    else if (stage == 4) hipLaunchKernelGGL(k_solve_wave, dim3(b.W), dim3(64), lds_bytes, s, b, sp, 1);
  }
  return VILO_OK;
}
