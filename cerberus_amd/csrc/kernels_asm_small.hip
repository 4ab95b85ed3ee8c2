// k_assemble_s: the camera-side normal equations of a window (what k_assemble_pose + k_assemble_bias of kernels_asm_full.hip build for larger batches: Ceres' Evaluate -> block-sparse
// J^T J behind estimator.cpp:1221-1236) for SMALL batches — up to one window per CU, where an iteration is a chain of kernel latencies and
// the assembly's is one chain of ~110 k cycles per window. Same owner-computes scatter (no atomics), cut by parallelism one window can use
// when it has a CU to itself (768 threads, 137 KB of LDS):
//   * the visual Gram slots of THREE chunks at a time: thread group g = tid / 256 takes the chunks g, g + 3, ... into its own copy of the
//     packed image (and of the gradient); the copies are added once in group order (the sums agree with a chunk-by-chunk
//     order to rounding, not bitwise — like the solver forms, the small-batch assembly is a form of its own);
//   * all IMU factor Grams of the window in LDS at once (62 KB over the second image copy and the staging areas, free by then): every
//     entry of A_kk / A_{k+1,k}^T / the coupling rows of every frame is independent work for 768 threads instead of a frame loop with
//     two barriers per frame;
//   * the output passes (tile image, prior rows, q) on twice the threads.
// The trust-region bookkeeping (accept_body.hpp) runs as its first phase and the second half of the frame-parallel visual form in extra
// workgroups.
#include <type_traits>
#include "solve_common.hpp"
#include "assemble_compact.hpp"
#include "accept_body.hpp"
#include "wave_common.hpp"
#include "lin_common.hpp"

using namespace vilo;

#define AS_NG 3           // thread groups of 256: chunks scattered at a time
#define AS_THREADS (256 * AS_NG)
// LDS map (doubles)
#define AS_CL0 0                         // [CL_N] packed image (group 0's copy, then the sum)
#define AS_R1 (AS_CL0 + CL_N)            // region with two lives:
#define AS_CL1 AS_R1                     //   [AS_NG - 1][CL_N] the other groups' copies of the image
#define AS_ST0 (AS_CL1 + (AS_NG - 1) * CL_N)   //   [AS_NG][AC_STAGE] the chunks the groups scatter; first: accept_body's scratch
#define AS_GR AS_R1                      //   [10 x 780] afterwards: the window's IMU factor Grams
#define AS_R1_N ((AS_NG - 1) * CL_N + AS_NG * AC_STAGE)
#define AS_GL (AS_R1 + AS_R1_N)          // [CD_N] gradient (group 0's, then the sum)
#define AS_GL1 (AS_GL + CD_N)            // [AS_NG - 1][CD_N] the other groups' shares of the gradient
#define AS_HD (AS_GL1 + (AS_NG - 1) * CD_N)   // [CD_N] diagonal
#define AS_VS (AS_HD + CD_N)             // [CD_N] v = g / dhat^2
#define AS_RT (AS_VS + CD_N)             // [12 x 9] rotation matrices of the frames, [11] = identity
#define AS_RED (AS_RT + 108)             // [48] per-wave partial sums
#define AS_TAB (AS_RED + 48)             // [64 unsigned = 32 doubles] chunk table
#define AS_PMAP (AS_TAB + 32)            // [CD_N shorts = 56 doubles] prior dimension of a camera dimension
#define AS_ACT (AS_PMAP + 56)            // [CD_N bytes = 28 doubles] activity of a camera dimension
#define AS_TOTAL (AS_ACT + 28)
static_assert(7800 <= AS_R1_N, "the Gram region lies inside the image copies + staging areas");
static_assert(256 + (AS_THREADS / 96) * 96 <= AS_NG * AC_STAGE, "accept_body's scratch fits the staging areas");
static_assert((AS_R1 & 1) == 0 && (AS_ST0 & 1) == 0 && (AC_STAGE & 1) == 0 && (CL_N & 1) == 0, "16-byte aligned areas");

__global__ void __launch_bounds__(AS_THREADS) __attribute__((amdgpu_waves_per_eu(AS_NG, AS_NG)))
k_assemble_s(BatchDev b, int jacobi_scaling, double min_lm_diagonal, double max_lm_diagonal, AcceptParams ap, int fuse_accept) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= b.W) {
    // the second half of the frame-parallel visual form, one packed wave per extra workgroup
    if (tid < 64) visual_reduce_body(b, (int)blockIdx.x - b.W, 1, true);
    return;
  }
  double *const Cl = lds + AS_CL0, *const Cl1 = lds + AS_CL1, *const gl = lds + AS_GL, *const gl1 = lds + AS_GL1, *const hd = lds + AS_HD, *const vS = lds + AS_VS;   // (Cl1 / gl1: AS_NG - 1 copies back to back)
  double *const Rt = lds + AS_RT, *const red = lds + AS_RED, *const grams = lds + AS_GR;
  unsigned *const chunk_tab = (unsigned *)(lds + AS_TAB);
  short *const inv_pmap = (short *)(lds + AS_PMAP);
  unsigned char *const act = (unsigned char *)(lds + AS_ACT);
  const int win = blockIdx.x;
  const long long c_k0 = pclk64();
  if (fuse_accept) {
    double *scr = lds + AS_ST0;
    accept_body(b, ap, scr, scr + 128, (int *)(scr + 128 + VILO_MAX_PRIOR_DIM), scr + 256);   // (scr + 256: AS_THREADS / 96 slices of the prior's H dx)
    __threadfence_block();
    __syncthreads();
  }
  const SolverState &st = b.st[win];
  if (st.done || !st.need_lin) return;
  const WinMeta wm = b.win[win];
  const int F = wm.n_frames, cmask = wm.const_mask, kb = wm.pad, pn = wm.prior_n;
  const double *gs = b.gram + (size_t)wm.gram_off * VILO_GRAMC;
  const double *igram = b.imu_gram + (size_t)win * 10 * 780;
  const double *pd = b.prior_dense + (size_t)win * PD_N;
  double *bimg = b.Bimg + (size_t)win * BI_N;
  const int grp = tid >> 8, lt = tid & 255;
  PCLK(if (tid == 0) { b.st[win].phase_clk[46] = clock64() - c_k0; b.st[win].phase_clk[36] = clock64(); });

  // ---- group 0's image starts from the prior's pre-assembled image (zeros without a prior), group 1's from zero ----
  {
    constexpr int NPV = (CL_N + AS_THREADS - 1) / AS_THREADS;
    double pv[NPV];
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      const int e = min(tid + AS_THREADS * u, CL_N - 1);
      int row = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      while (((row + 1) * (row + 2)) / 2 <= e) ++row;
      while ((row * (row + 1)) / 2 > e) --row;
      pv[u] = pd[PD_C + row * PD_CLD + (e - (row * (row + 1)) / 2)];
    }
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
      const int e = tid + AS_THREADS * u;
      if (e < CL_N) {
        Cl[e] = pv[u];
#pragma unroll
        for (int g2 = 0; g2 < AS_NG - 1; ++g2) Cl1[g2 * CL_N + e] = 0.0;
      }
    }
  }
  for (int e = tid; e < CD_N; e += AS_THREADS) {
    inv_pmap[e] = -1; act[e] = cd_active(e, F, cmask) ? 1 : 0;
#pragma unroll
    for (int g2 = 0; g2 < AS_NG - 1; ++g2) gl1[g2 * CD_N + e] = 0.0;
  }
  if (tid < 11) {
    const m3 R = qR(ldq_pose(b.x + (size_t)win * XSTRIDE + XO_POSE + 7 * tid));   // (the accepted state = the point the slots were linearised at)
#pragma unroll
    for (int q = 0; q < 9; ++q) Rt[9 * tid + q] = R.a[q];
  } else if (tid < 20) {
    Rt[99 + (tid - 11)] = ((tid - 11) % 4 == 0) ? 1.0 : 0.0;
  }
  if (tid < min(wm.n_chunks, 64)) {
    const ChunkMeta cm = b.chunk[wm.chunk_off + tid];
    chunk_tab[tid] = (unsigned)cm.s | ((unsigned)cm.kmax << 8) | ((unsigned)(cm.gram_off - wm.gram_off) << 16);
  }
  __syncthreads();
  if (tid < pn) inv_pmap[b.prior_map[(size_t)win * 96 + tid]] = (short)tid;
  __syncthreads();
  // gradient starts from the prior's b0 + H dx (H dx at the current point was formed by the bookkeeping when it evaluated this point's cost)
  for (int e = tid; e < CD_N; e += AS_THREADS) {
    const int pi = inv_pmap[e];
    gl[e] = (pn > 0 && pi >= 0) ? b.prior_b0[(size_t)win * 96 + pi] + b.prior_hd[(size_t)win * 96 + pi] : 0.0;
  }
  __syncthreads();

  PCLK(if (tid == 0) b.st[win].phase_clk[37] = clock64());
  // ---- visual Gram slots: AS_NG chunks per trip, one per thread group, each into its own image / gradient copy ----
  {
    double *const Cg = grp ? Cl1 + (grp - 1) * CL_N : Cl, *const gg = grp ? gl1 + (grp - 1) * CD_N : gl, *const stage = lds + AS_ST0 + grp * AC_STAGE;
    auto rmw = [&](int hi, int lo, double v) { lds_add(&Cg[cl_pos(hi, lo)], v); };   // hi >= lo
    const int nch = min(wm.n_chunks, 64), ntrip = (nch + AS_NG - 1) / AS_NG;
    double pf[8];
    auto prefetch = [&](int ch) {
      if (ch < nch) {
        const unsigned ct = chunk_tab[ch];
        const int n = (int)((ct >> 8) & 255) * VILO_GRAMC;
        const double *src = gs + (size_t)(ct >> 16) * VILO_GRAMC;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int e = lt + 256 * i; pf[i] = (e < n) ? src[e] : 0.0; }
      }
    };
    prefetch(grp);
    for (int trip = 0; trip < ntrip; ++trip) {
      const int ch = AS_NG * trip + grp;
      lds_barrier();   // (the previous trip's readers are done)
      if (ch < nch) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int e = lt + 256 * i; if (e < AC_STAGE) stage[e] = pf[i]; }
      }
      lds_barrier();
      prefetch(ch + AS_NG);
      if (ch < nch) {
        const unsigned ct = chunk_tab[ch];
        const int cs_ = (int)(ct & 255), km = (int)((ct >> 8) & 255);
        assemble_visual_compact_chunk(lt, cs_, km, stage, Rt, rmw, [&](int cd, double v) { lds_add(&gg[cd], v); });
        if (lt >= 64 && lt < 128) {
          // wave 1 of the group: the {tic, tic2}^2 entries, three lanes per entry (each a third of the chunk's frames), partial sums added in lane order
          const int wl = lt - 64, q = wl % 21, g3 = min(wl / 21, 2);
          const double part = (wl < 63) ? ac_t8_partial(q, g3, 3, cs_, km, stage, Rt) : 0.0;
          const double p1 = __shfl(part, q + 21, 64), p2 = __shfl(part, q + 42, 64);
          if (wl < 21) ac_t8_apply(q, (part + p1) + p2, rmw);
        } else if (lt >= 128 && lt < 192) {
          // wave 2 of the group: the tic / tic2 x pose entries, two lanes per entry (odd / even frames), the sums over the frames added in lane order
          const int wl = lt - 128, q = wl % 18, par = min(wl / 18, 1);
          double s5 = 0.0, s6 = 0.0;
          if (wl < 36) ac_t56_partial(q, par, cs_, km, stage, Rt, rmw, s5, s6);
          const double o5 = __shfl(s5, q + 18, 64), o6 = __shfl(s6, q + 18, 64);
          if (wl < 18) ac_t56_apply(q, cs_, s5 + o5, s6 + o6, rmw);
        }
      }
    }
  }
  __syncthreads();
  // the copies become one (the groups' sums added in group order)
  for (int e = tid; e < CL_N; e += AS_THREADS) {
    double sacc = Cl[e];
#pragma unroll
    for (int g2 = 0; g2 < AS_NG - 1; ++g2) sacc += Cl1[g2 * CL_N + e];
    Cl[e] = sacc;
  }
  for (int e = tid; e < CD_N; e += AS_THREADS) {
    double sacc = gl[e];
#pragma unroll
    for (int g2 = 0; g2 < AS_NG - 1; ++g2) sacc += gl1[g2 * CD_N + e];
    gl[e] = sacc;
  }
  __syncthreads();

  PCLK(if (tid == 0) b.st[win].phase_clk[38] = clock64());
  // ---- IMU factors: all of the window's packed 39 x 39 Grams [pose_i 6 | speed / leg-bias_i 13 | pose_j 6 | speed / leg-bias_j 13 | r]
  //      (780 entries each) come into LDS with coalesced loads, all in flight; then every entry that goes to the solver is independent.
  //      Per frame k: Gi = factor k (frame k is its "i"), Gj = factor k - 1.
  //        [0, 169)    A_kk             = prior (frame kb) + Gi[6 + .][6 + .] + Gj[25 + .][25 + .]
  //        [169, 338)  A_{k+1,k}^T      = Gi[6 + j][25 + i]
  //        [338, 626)  coupling rows    with poses k - 1 (Gj), k (Gi + Gj), k + 1 (Gi); rows 13 .. 15 zero padding ----
  {
    const int ng = max(F - 1, 0) * 780;
    constexpr int NGP = (7800 + AS_THREADS - 1) / AS_THREADS;
    double gp[NGP];
#pragma unroll
    for (int u = 0; u < NGP; ++u) { const int e = tid + AS_THREADS * u; gp[u] = (e < ng) ? igram[e] : 0.0; }
#pragma unroll
    for (int u = 0; u < NGP; ++u) { const int e = tid + AS_THREADS * u; if (e < 7800) grams[e] = gp[u]; }
  }
  __syncthreads();
  {
    // (three passes, one per kind of block, so that a wave runs one body; the activity of a speed / leg-bias dimension is arithmetic:
    // frame inside the window and not a constant leg-bias block; the LDS reads of a pass are independent and go out together)
    const bool lb_off = (cmask & CONST_LB) != 0;
    auto on = [&](int k, int i) { return k < F && !(i >= 9 && lb_off); };
    // A_kk
#pragma unroll
    for (int it = 0; it < (11 * 169 + AS_THREADS - 1) / AS_THREADS; ++it) {
      const int e = tid + AS_THREADS * it;
      if (e < F * 169) {
        const int k = e / 169, u = e - 169 * k, i = u / 13, j = u - 13 * i;
        const bool has_i = k < F - 1, has_j = k >= 1;
        const double gi = has_i ? grams[780 * k + tri39(6 + min(i, j), 6 + max(i, j))] : 0.0;
        const double gj = has_j ? grams[780 * (k - 1) + tri39(25 + min(i, j), 25 + max(i, j))] : 0.0;
        double v;
        if (!on(k, i) || !on(k, j)) v = (i == j) ? 1.0 : 0.0;
        else {
          v = (k == kb) ? pd[PD_AD + 169 * kb + u] : 0.0;
          if (has_i) v += gi;
          if (has_j) v += gj;
        }
        bimg[BI_AD + e] = v;
      }
    }
    // A_{k+1,k}^T of the factors' frames (a partial window's remaining blocks are zeroed below)
#pragma unroll
    for (int it = 0; it < (10 * 169 + AS_THREADS - 1) / AS_THREADS; ++it) {
      const int e = tid + AS_THREADS * it;
      if (e < (F - 1) * 169) {
        const int k = e / 169, ji = e - 169 * k, j = ji / 13, i = ji - 13 * j;   // i: dimension of frame k + 1, j: of frame k
        const double gi = grams[780 * k + tri39(6 + j, 25 + i)];
        bimg[BI_AOT + e] = (on(k + 1, i) && on(k, j)) ? gi : 0.0;
      }
    }
    // coupling rows with poses k - 1 (Gj), k (Gi + Gj), k + 1 (Gi); rows 13 .. 15 zero padding
#pragma unroll
    for (int it = 0; it < (11 * 288 + AS_THREADS - 1) / AS_THREADS; ++it) {
      const int e = tid + AS_THREADS * it;
      if (e < F * 288) {
        const int k = e / 288, is = e - 288 * k, i = is / 18, sx = is - 18 * i, df = sx / 6, c = sx - 6 * df;
        const bool has_i = k < F - 1, has_j = k >= 1;
        const int f = k - 1 + df, i12 = min(i, 12);
        const double *Gi = grams + 780 * k, *Gj = grams + 780 * (k - 1);
        // (one Gi and one Gj entry per kind of column block; which exist depends on df)
        const int ei = df == 1 ? tri39(c, 6 + i12) : tri39(6 + i12, 19 + c), ej = df == 1 ? tri39(19 + c, 25 + i12) : tri39(c, 25 + i12);
        const double gi = (has_i && df >= 1) ? Gi[ei] : 0.0, gj = (has_j && df <= 1) ? Gj[ej] : 0.0;
        double v = 0.0;
        if (i < 13 && f >= 0 && f < F && on(k, i12)) {
          if (has_i && df >= 1) v += gi;
          if (has_j && df <= 1) v += gj;
        }
        bimg[BI_BS + e] = v;
      }
    }
  }
  // pose blocks of the factors -> the pose image: I1 pose_i x pose_i (21, twin + 19), I3 pose gradient (6, twin + 19), I4 pose_i x pose_j (36).
  // Consecutive factors meet in the diagonal block of the frame they share: one owner thread per entry walks the factors in order.
  if (tid >= AS_THREADS - 64 && tid < AS_THREADS - 1) {
    int pa = 0, pbc = 0, pcls = 0;
    const int q = tid - (AS_THREADS - 64);
    if (q < 21) { pcls = 1; int rem = q; while (rem >= 6 - pa) { rem -= 6 - pa; ++pa; } pbc = pa + rem; }
    else if (q < 27) { pcls = 3; pa = q - 21; pbc = 38; }
    else { pcls = 4; pa = (q - 27) / 6; pbc = 19 + (q - 27) % 6; }
    const int pe1 = tri39(pa, pbc), pe2 = (pcls == 4) ? pe1 : tri39(pa + 19, pcls == 3 ? 38 : pbc + 19);
    for (int k = 0; k + 1 < F; ++k) {
      const double vm = grams[780 * k + pe1], vt = grams[780 * k + pe2];
      if (pcls == 1) { lds_add(&Cl[cl_pos(6 * k + pbc, 6 * k + pa)], vm); lds_add(&Cl[cl_pos(6 * (k + 1) + pbc, 6 * (k + 1) + pa)], vt); }
      else if (pcls == 3) { lds_add(&gl[6 * k + pa], vm); lds_add(&gl[6 * (k + 1) + pa], vt); }
      else lds_add(&Cl[cl_pos(6 * (k + 1) + (pbc - 19), 6 * k + pa)], vm);
    }
  }
  // diagonal and gradient of the frames' speed / leg-bias dimensions (the prior's share of the gradient is there already)
  if (tid < 13 * F) {
    const int k = tid / 13, i = tid - 13 * k, cd = CD_B0 + 13 * k + i;
    const bool has_i = k < F - 1, has_j = k >= 1;
    const double *Gi = grams + 780 * k, *Gj = grams + 780 * (k - 1);
    double h = 1.0, g = gl[cd];
    if (act[cd]) {
      h = (k == kb) ? pd[PD_AD + 169 * kb + i * 14] : 0.0;
      if (has_i) { h += Gi[tri39(6 + i, 6 + i)]; g += Gi[tri39(6 + i, 38)]; }
      if (has_j) { h += Gj[tri39(25 + i, 25 + i)]; g += Gj[tri39(25 + i, 38)]; }
    }
    hd[cd] = h;
    gl[cd] = g;
  }
  __syncthreads();
  PCLK(if (tid == 0) b.st[win].phase_clk[39] = clock64());
  // ---- diagonal of the pose part; gradient of all 224 camera dimensions (inactive ones zero) ----
  for (int cd = tid; cd < CD_N; cd += AS_THREADS) {
    double g = gl[cd];
    if (cd < CD_B0) hd[cd] = act[cd] ? Cl[cl_pos(cd, cd)] : 1.0;
    else if (cd >= CD_B0 + 13 * F) hd[cd] = 1.0;   // (frames beyond the window, padding)
    if (!act[cd]) g = 0.0;
    gl[cd] = g;
    b.cam_gin[(size_t)win * CD_N + cd] = g;
  }
  __syncthreads();
  PCLK(if (tid == 0) b.st[win].phase_clk[40] = clock64());
  // ---- Jacobi scaling 1 / (1 + sqrt(H_ii)) frozen at the first linearisation, dogleg diagonal clamp(diag, 1e-6, 1e32) in the scaled
  //      space, v = D^-2 g (Ceres 1.14 TrustRegionMinimizer / DoglegStrategy) ----
  double part_gn = 0.0, part_gmax = 0.0, part_q = 0.0;
  for (int cd = tid; cd < CD_N; cd += AS_THREADS) {
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
  for (int idx = tid; idx < CIMG_N; idx += AS_THREADS) {
    const int t = idx >> 8, rr = (idx >> 4) & 15, cc = idx & 15;
    const int row = 16 * c_tI[t] + rr, col = 16 * c_tJ[t] + cc;
    const double v = (act[row] && act[col]) ? Cl[cl_pos(max(row, col), min(row, col))] : (row == col ? 1.0 : 0.0);
    b.Cimg[(size_t)win * CIMG_N + idx] = v;
    part_q += ((c_tI[t] == c_tJ[t]) ? 1.0 : 2.0) * vS[row] * v * vS[col];   // (a diagonal tile holds both triangles)
  }
  PCLK(if (tid == 0) b.st[win].phase_clk[42] = clock64());
  // ---- prior rows of the frame whose speed / leg-bias block the prior touches (rows 13 .. 15: zero padding) ----
  for (int e = tid; e < 1280; e += AS_THREADS) {
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
    for (int e = tid + 169 * F; e < 11 * 169; e += AS_THREADS) { const int ij = e % 169; bimg[BI_AD + e] = (ij / 13 == ij % 13) ? 1.0 : 0.0; }
    for (int e = tid + 169 * max(F - 1, 0); e < 10 * 169; e += AS_THREADS) bimg[BI_AOT + e] = 0.0;
    for (int e = tid + 288 * F; e < 11 * 288; e += AS_THREADS) bimg[BI_BS + e] = 0.0;
  }
  PCLK(if (tid == 0) b.st[win].phase_clk[43] = clock64());
  // ---- q = v^T H v of the speed / leg-bias rows: the blocks just written come back (coalesced, L2-resident; all loads of a kind in flight) ----
  __threadfence_block();   // (this workgroup's stores above must be visible to its loads below)
  __syncthreads();
  {
    constexpr int NV = (11 * 169 + AS_THREADS - 1) / AS_THREADS;
    double val[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) val[u] = bimg[BI_AD + min(tid + AS_THREADS * u, 11 * 169 - 1)];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = tid + AS_THREADS * u;
      if (e < 11 * 169) {
        const int k = e / 169, ij = e - 169 * k, i = ij / 13, j = ij - 13 * i;
        part_q += vS[CD_B0 + 13 * k + i] * val[u] * vS[CD_B0 + 13 * k + j];
      }
    }
  }
  {
    constexpr int NV = (10 * 169 + AS_THREADS - 1) / AS_THREADS;
    double val[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) val[u] = bimg[BI_AOT + min(tid + AS_THREADS * u, 10 * 169 - 1)];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = tid + AS_THREADS * u;
      if (e < 10 * 169) {
        const int k = e / 169, ji = e - 169 * k, j = ji / 13, i = ji - 13 * j;
        part_q += 2.0 * vS[CD_B0 + 13 * (k + 1) + i] * val[u] * vS[CD_B0 + 13 * k + j];
      }
    }
  }
  {
    constexpr int NV = (11 * 288 + AS_THREADS - 1) / AS_THREADS;
    double val[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) val[u] = bimg[BI_BS + min(tid + AS_THREADS * u, 11 * 288 - 1)];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = tid + AS_THREADS * u;
      if (e < 11 * 288) {
        const int k = e / 288, is = e - 288 * k, i = is / 18, sx = is - 18 * i, f = k - 1 + sx / 6, c = sx % 6;
        if (val[u] != 0.0) part_q += 2.0 * vS[CD_B0 + 13 * k + min(i, 12)] * val[u] * vS[min(max(6 * f + c, 0), 79)];
      }
    }
  }
  PCLK(if (tid == 0) b.st[win].phase_clk[44] = clock64());
  // camera-side sums of |D^-1 g|^2, max |g| and q (the landmarks add theirs in the solver): waves in fixed order
  part_q = wave_sum(part_q); part_gn = wave_sum(part_gn); part_gmax = wave_max(part_gmax);
  constexpr int NWV = AS_THREADS / 64;
  if ((tid & 63) == 0) { red[tid >> 6] = part_q; red[NWV + (tid >> 6)] = part_gn; red[2 * NWV + (tid >> 6)] = part_gmax; }
  __syncthreads();
  PCLK(if (tid == 0) b.st[win].phase_clk[45] = clock64());
  if (tid == 0) {
    double sq_ = 0.0, sg_ = 0.0, mx = 0.0;
    for (int w = 0; w < NWV; ++w) { sq_ += red[w]; sg_ += red[NWV + w]; mx = fmax(mx, red[2 * NWV + w]); }
    bimg[BI_SCAL + 0] = sq_;
    bimg[BI_SCAL + 1] = sg_;
    bimg[BI_SCAL + 2] = mx;
  }
}

// k_assemble_s for batches of up to VILO_ASM_SMALL_MAX_WINDOWS windows with compact slots (reduce_waves: extra workgroups that finish the
// frame-parallel visual form).
int vilo_assemble_small_max() {
  // (one window per CU: measured 256 windows + 4 %, 384 - 8 % against three workgroups per CU)
  static const int small_max = [] { const char *e = getenv("VILO_ASM_SMALL_MAX_WINDOWS"); return e ? atoi(e) : 256; }();
  return small_max;
}
bool vilo_assemble_small_takes(const BatchDev &b) { return b.compact && !b.full_regime && b.W <= vilo_assemble_small_max(); }
int vilo_launch_assemble_small(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s, const AcceptParams *ap, int reduce_waves) {
  const size_t lds_bytes = (size_t)AS_TOTAL * sizeof(double);
  if (!ctx->asm_s_attr_set) {
    if (hipFuncSetAttribute((const void *)k_assemble_s, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) { ctx->err = "k_assemble_s: dynamic LDS opt-in failed"; return VILO_ERR_HIP; }
    ctx->asm_s_attr_set = true;
  }
  hipLaunchKernelGGL(k_assemble_s, dim3(b.W + reduce_waves), dim3(AS_THREADS), lds_bytes, s, b, sp.jacobi_scaling, sp.min_lm_diagonal, sp.max_lm_diagonal, *ap, 1);
  return 1;
}
