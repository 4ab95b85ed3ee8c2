// Marginalisation half of Estimator::optimization() (estimator.cpp:1247-1455) and MarginalizationInfo
// (marginalization_factor.cpp:98-333) on gfx950, plus double2vector's gauge fix (estimator.cpp:903-957).
//
// The factors touching the dropped blocks are linearised by the same kernels the solver uses
// (k_visual_linearize / k_imu_linearize: residuals, Jacobians, Huber corrector), then one workgroup per window
//   * scatters J^T J / J^T r into A, b ordered [dropped | kept] (ThreadsConstructA, :150-181, without the 4 zero-
//     initialised thread-local copies),
//   * eigen-decomposes Amm by a parallel cyclic Jacobi method and forms the eps-thresholded pseudo-inverse (:281-286),
//   * forms the Schur complement A' = Arr - Arm Amm^+ Amr, b' = brr - Arm Amm^+ bmm (:289-295),
//   * eigen-decomposes A' and emits J0 = sqrt(S) V^T, r0 = sqrt(S^-1) V^T b' (:297-305).
// Address-keyed bookkeeping (addr_shift, estimator.cpp:1358-1370) is replaced by integer block ids.
#include <algorithm>
#include <chrono>

#include <type_traits>
#include "solve_common.hpp"

using namespace vilo;

int vilo_marg_linearize(vilo_ctx *ctx, BatchDev &b);   // kernels_solve.hip

namespace {

#define MT 256

__device__ double blk_sum(double v, double *red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = MT / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// Parallel two-sided cyclic Jacobi: A (n x n symmetric, row-major, ld = n, global/L2) -> diag(A) = eigenvalues,
// V columns = eigenvectors. Round-robin pairing gives n/2 disjoint rotations per step; row phase then column phase.
__device__ void jacobi_eigh_dev(double *A, double *V, int n, double *cs /*LDS 2*(n/2+1)*/, int *pq /*LDS 2*(n/2+1)*/, double *red) {
  const int tid = threadIdx.x;
  for (int e = tid; e < n * n; e += MT) V[e] = (e / n == e % n) ? 1.0 : 0.0;
  __syncthreads();
  const int ne = (n + 1) & ~1, half = ne / 2;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, dg = 0.0;
    for (int e = tid; e < n * n; e += MT) {
      const int i = e / n, j = e % n;
      const double v = A[e];
      if (i == j) dg += v * v; else if (j > i) off += v * v;
    }
    off = blk_sum(off, red);
    dg = blk_sum(dg, red);
    if (off <= 1e-60 || off <= 1e-32 * dg) break;
    for (int step = 0; step < ne - 1; ++step) {
      if (tid < half) {
        // circle method: position tid plays position ne-1-tid; player at position k is (k == ne-1) ? ne-1 : (k + step) % (ne-1)
        const int ka = tid, kb = ne - 1 - tid;
        int p = (ka == ne - 1) ? ne - 1 : (ka + step) % (ne - 1);
        int q = (kb == ne - 1) ? ne - 1 : (kb + step) % (ne - 1);
        if (p > q) { const int t_ = p; p = q; q = t_; }
        double c = 1.0, s = 0.0;
        if (q < n) {
          const double apq = A[p * n + q];
          if (apq != 0.0) {
            const double app = A[p * n + p], aqq = A[q * n + q];
            const double tau = (aqq - app) / (2.0 * apq);
            const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + t * t);
            s = t * c;
          }
        } else {
          q = -1;
        }
        pq[2 * tid] = p; pq[2 * tid + 1] = q;
        cs[2 * tid] = c; cs[2 * tid + 1] = s;
      }
      __syncthreads();
      // rows: A <- J^T A
      for (int e = tid; e < half * n; e += MT) {
        const int r = e / n, k = e % n;
        const int p = pq[2 * r], q = pq[2 * r + 1];
        if (q < 0) continue;
        const double c = cs[2 * r], s = cs[2 * r + 1];
        const double apk = A[p * n + k], aqk = A[q * n + k];
        A[p * n + k] = c * apk - s * aqk;
        A[q * n + k] = s * apk + c * aqk;
      }
      __syncthreads();
      // columns: A <- A J, V <- V J
      for (int e = tid; e < half * n; e += MT) {
        const int r = e / n, k = e % n;
        const int p = pq[2 * r], q = pq[2 * r + 1];
        if (q < 0) continue;
        const double c = cs[2 * r], s = cs[2 * r + 1];
        const double akp = A[k * n + p], akq = A[k * n + q];
        A[k * n + p] = c * akp - s * akq;
        A[k * n + q] = s * akp + c * akq;
        const double vkp = V[k * n + p], vkq = V[k * n + q];
        V[k * n + p] = c * vkp - s * vkq;
        V[k * n + q] = s * vkp + c * vkq;
      }
      __syncthreads();
    }
  }
  __syncthreads();
}

struct MargWin {
  int m, n;             // dropped / kept local dims
  int n_drop_lm;        // landmarks among the dropped dims (they follow the 19 frame-0 dims; mode 0)
  int mode;
  int has_imu, prior_n;
  int general;          // 1: this window goes through the global-memory eigen path (set by the host from k_marginalize_lds' verdict)
  int pad1;
  long long scratch_off;  // doubles into the scratch arena
  int cdmap[CD_N];      // camera dim -> index in [dropped | kept] ordering, -1 if absent
};

// scratch layout per window (doubles): A (T*T) | b (T) | Amm (m*m) | Vm (m*m) | Ainv (m*m) | tmp (n*m) | Ar (n*n) | V2 (n*n) | br (n)
__global__ void __launch_bounds__(MT) k_marginalize(BatchDev bd, const MargWin *mw, const int *drop_lm /*[W][maxL0] device-order local idx*/,
                                                    int max_l0, double *scratch, double *J0_out, double *r0_out, int *status) {
  __shared__ double red[MT];
  __shared__ double cs[2 * 600];
  __shared__ int pq[2 * 600];
  __shared__ double dx[VILO_MAX_PRIOR_DIM];
  const int win = blockIdx.x, tid = threadIdx.x;
  const MargWin &M = mw[win];
  const WinMeta wm = bd.win[win];
  const int m = M.m, n = M.n, T = m + n;
  if (m == 0 || n == 0 || !M.general) return;
  double *A = scratch + M.scratch_off, *bv = A + (size_t)T * T, *Amm = bv + T, *Vm = Amm + (size_t)m * m, *Ainv = Vm + (size_t)m * m;
  double *tmp = Ainv + (size_t)m * m, *Ar = tmp + (size_t)n * m, *V2 = Ar + (size_t)n * n, *br = V2 + (size_t)n * n;
  const double *x = bd.x + (size_t)win * XSTRIDE;
  for (size_t e = tid; e < (size_t)T * T + T; e += MT) A[e] = 0.0;
  __syncthreads();
  auto add = [&](int ci, int cj, double v) {   // camera dims -> A
    const int i = M.cdmap[ci], j = M.cdmap[cj];
    if (i >= 0 && j >= 0) A[(size_t)i * T + j] += v;
  };
  // ---- prior factor: J^T J = H_prior, J^T r = b0 + H_prior dx ----
  if (M.prior_n > 0) {
    const int pn = M.prior_n;
    const double *Hp = bd.prior_H + (size_t)win * 96 * 96, *b0 = bd.prior_b0 + (size_t)win * 96;
    const int *pmap = bd.prior_map + (size_t)win * 96;
    if (tid < wm.prior_nb)
      prior_dx(x + bd.prior_bstate[win * 40 + tid], bd.prior_x0 + (size_t)win * 280 + bd.prior_bxoff[win * 40 + tid],
               bd.prior_bsize[win * 40 + tid], dx + bd.prior_bidx[win * 40 + tid]);
    __syncthreads();
    for (int e = tid; e < pn * pn; e += MT) add(pmap[e / pn], pmap[e % pn], Hp[e]);
    for (int i = tid; i < pn; i += MT) {
      double s = b0[i];
      for (int q = 0; q < pn; ++q) s += Hp[(size_t)q * pn + i] * dx[q];
      const int t = M.cdmap[pmap[i]];
      if (t >= 0) bv[t] += s;
    }
    __syncthreads();
  }
  if (M.mode == 0) {
    // ---- IMULegFactor(0, 1) ----
    if (M.has_imu) {
      const double *lin = bd.imu_lin + (size_t)win * 10 * 31 * 39;
      for (int e = tid; e < 39 * 39; e += MT) {
        const int a = e / 39, c = e % 39;
        if (a == 38) continue;
        double s = 0.0;
        for (int i = 0; i < 31; ++i) s += lin[i * 39 + a] * lin[i * 39 + c];
        auto cdof = [](int cc) { return cc < 6 ? cc : (cc < 19 ? CD_B0 + (cc - 6) : (cc < 25 ? 6 + (cc - 19) : CD_B0 + 13 + (cc - 25))); };
        if (c == 38) { const int t = M.cdmap[cdof(a)]; if (t >= 0) bv[t] += s; }
        else add(cdof(a), cdof(c), s);
      }
      __syncthreads();
    }
    // ---- visual factors of the landmarks that start in frame 0: camera-side Gram slots of the s = 0 chunks ----
    for (int ch = 0; ch < wm.n_chunks; ++ch) {
      const ChunkMeta cm = bd.chunk[wm.chunk_off + ch];
      if (cm.s != 0) continue;
      for (int t = 0; t < cm.kmax; ++t) {
        const double *gs = bd.gram + (size_t)(cm.gram_off + t) * VILO_GRAM;
        for (int e = tid; e < VILO_GRAM26; e += MT) {   // the 26-column view of the slot
          int a = 0, rem = e;
          while (rem >= 26 - a) { rem -= 26 - a; ++a; }
          const int bc = a + rem;
          if (t == 0 && ((a >= 6 && a < 12) || (bc >= 6 && bc < 12))) continue;
          auto cdof = [&](int c) { return c < 6 ? c : (c < 12 ? 6 * t + (c - 6) : (c < 18 ? CD_EX0 + c - 12 : (c < 24 ? CD_EX1 + c - 18 : CD_TD))); };
          double sg;
          const int ge = gram26_index(a, bc, sg);
          const double v = sg * gs[ge];
          if (bc == 25) { if (a < 25) { const int q = M.cdmap[cdof(a)]; if (q >= 0) bv[q] += v; } }
          else {
            add(cdof(a), cdof(bc), v);
            if (cdof(a) != cdof(bc)) add(cdof(bc), cdof(a), v);
          }
        }
        __syncthreads();
      }
    }
    // landmark (inverse depth) rows / columns
    const double *wl = bd.lm_w + 80 * (size_t)wm.lm_off;
    for (int e = tid; e < M.n_drop_lm * 80; e += MT) {
      const int li = e / 80, a = e % 80;
      const int l = drop_lm[(size_t)win * max_l0 + li];
      const int row = M.m - M.n_drop_lm + li;   // landmarks follow the dropped frame-0 dims (19 with leg biases, 15 without)
      if (a == 79) {
        A[(size_t)row * T + row] += bd.lm_E[wm.lm_off + l];
        bv[row] += bd.lm_gbuf[0][wm.lm_off + l];
      } else {
        const int t = M.cdmap[a];
        const double v = wl[(size_t)a * wm.L + l];
        if (t >= 0) { A[(size_t)row * T + t] += v; A[(size_t)t * T + row] += v; }
      }
    }
    __syncthreads();
  }
  // ---- Amm = 1/2 (Amm + Amm^T), eigen, pseudo-inverse (eps = 1e-8) ----
  const double eps = 1e-8;
  for (int e = tid; e < m * m; e += MT) {
    const int i = e / m, j = e % m;
    Amm[e] = 0.5 * (A[(size_t)i * T + j] + A[(size_t)j * T + i]);
  }
  __syncthreads();
  jacobi_eigh_dev(Amm, Vm, m, cs, pq, red);
  for (int e = tid; e < m * m; e += MT) {
    const int i = e / m, j = e % m;
    double s = 0.0;
    for (int k = 0; k < m; ++k) {
      const double w = Amm[(size_t)k * m + k];
      if (w > eps) s += Vm[(size_t)i * m + k] * (1.0 / w) * Vm[(size_t)j * m + k];
    }
    Ainv[e] = s;
  }
  __syncthreads();
  // tmp = Arm Amm^+ ; A' = Arr - tmp Amr ; b' = brr - tmp bmm
  for (int e = tid; e < n * m; e += MT) {
    const int i = e / m, j = e % m;
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += A[(size_t)(m + i) * T + k] * Ainv[(size_t)k * m + j];
    tmp[e] = s;
  }
  __syncthreads();
  for (int e = tid; e < n * n; e += MT) {
    const int i = e / n, j = e % n;
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += tmp[(size_t)i * m + k] * A[(size_t)k * T + m + j];
    Ar[e] = A[(size_t)(m + i) * T + m + j] - s;
  }
  for (int i = tid; i < n; i += MT) {
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += tmp[(size_t)i * m + k] * bv[k];
    br[i] = bv[m + i] - s;
  }
  __syncthreads();
  // SelfAdjointEigenSolver reads the lower triangle: symmetrise from it
  for (int e = tid; e < n * n; e += MT) {
    const int i = e / n, j = e % n;
    if (j > i) Ar[e] = Ar[(size_t)j * n + i];
  }
  __syncthreads();
  // keep A' for the caller's invariants before it is diagonalised: stored after br
  double *Akeep = br + n;
  for (int e = tid; e < n * n; e += MT) Akeep[e] = Ar[e];
  __syncthreads();
  jacobi_eigh_dev(Ar, V2, n, cs, pq, red);
  double *J0 = J0_out + (size_t)win * VILO_MAX_PRIOR_DIM * VILO_MAX_PRIOR_DIM, *r0 = r0_out + (size_t)win * VILO_MAX_PRIOR_DIM;
  for (int e = tid; e < n * n; e += MT) {
    const int i = e / n, j = e % n;
    const double S = Ar[(size_t)i * n + i];
    J0[e] = (S > eps) ? sqrt(S) * V2[(size_t)j * n + i] : 0.0;
  }
  for (int i = tid; i < n; i += MT) {
    const double S = Ar[(size_t)i * n + i];
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += V2[(size_t)j * n + i] * br[j];
    r0[i] = (S > eps) ? sqrt(1.0 / S) * s : 0.0;
    if (!isfinite(r0[i])) status[win] = 1;
  }
}

// double2vector gauge fix + re-pack (estimator.cpp:903-957, 848-873)
// ---------------------------------------------------------------------------------------------------------------------
// LDS-resident marginalisation (the path every well-conditioned window takes).
//
// MarginalizationInfo::marginalize inverts Amm through an eigen-decomposition with eigenvalues <= eps = 1e-8 dropped
// (marginalization_factor.cpp:281-286). When every eigenvalue of Amm exceeds eps that pseudo-inverse IS the inverse, and the
// Schur complement can be formed by elimination in the order the block structure suggests: the dropped landmarks first (their
// block of Amm is diagonal: an inverse depth only couples to camera-side blocks), then the <= 19 dense dims of frame 0 by a
// Cholesky factorisation. The kernel certifies "lambda_min(Amm) > eps" by factorising Amm - eps I the same way (all pivots
// positive) and hands the window to k_marginalize (global-memory Jacobi on the full Amm) when the certificate fails.
// The second decomposition (A' -> J0 = sqrt(S) V^T, marginalization_factor.cpp:297-305) needs a rank decision (a first prior carries
// the unobservable gauge directions) and the products sqrt(S) V^T, S^-1/2 V^T b, not V itself: prior_factor_lds below.
#define MGT 1024
constexpr int MG_NMAX = VILO_MAX_PRIOR_DIM;       // 96
constexpr int MG_LD = MG_NMAX + 1;                // odd leading dimension: column walks spread over the LDS banks
constexpr int MG_TMAX = 19 + MG_NMAX;             // dense dropped dims + kept dims
constexpr int MG_TILE = 32;                       // landmarks per elimination tile
constexpr int MG_R0 = 2 * MG_NMAX * MG_LD;        // 18624 doubles: A1 (TMAX^2 = 13225) + tile (TMAX * 32), later A' | V
constexpr int MG_TILE_OFF = 13312;
constexpr int MG_SMALL = MG_TMAX + 3 * MG_TILE + 19 * 19 + 2 * 48 + 19 + MG_NMAX + 8;
constexpr int MG_LDS_DOUBLES = MG_R0 + MG_SMALL;
static_assert(MG_TILE_OFF >= MG_TMAX * MG_TMAX && MG_TILE_OFF + MG_TMAX * MG_TILE <= MG_R0, "LDS plan");
static_assert(MG_LDS_DOUBLES * 8 + 2 * 80 * 4 + 96 * 8 + 2 * 96 * 8 + 64 <= 160 * 1024, "LDS budget");   // dynamic + act_a / act_t, dx, dgb, flags

// in-place Cholesky of the leading d x d block (lower triangle) of M; all threads call. *fail is set when a pivot is not positive.
__device__ void chol_lds(double *M, int ld, int d, int *fail) {
  const int tid = threadIdx.x;
  for (int k = 0; k < d; ++k) {
    if (tid == 0) {
      const double p = M[k * ld + k];
      if (!(p > 0.0)) { *fail = 1; M[k * ld + k] = 1.0; } else M[k * ld + k] = sqrt(p);
    }
    __syncthreads();
    if (tid > k && tid < d) M[tid * ld + k] /= M[k * ld + k];
    __syncthreads();
    for (int e = tid; e < d * d; e += MGT) {
      const int i = e / d, j = e % d;
      if (j > k && i >= j) M[i * ld + j] -= M[i * ld + k] * M[j * ld + k];
    }
    __syncthreads();
  }
}

// A' -> J0, r0 (marginalization_factor.cpp:297-305: J0 = sqrt(S) V^T, r0 = S^-1/2 V^T b over the eigenpairs with S > eps) without forming V.
//
// With A' = X X^T, rotating the columns of X from the right until they are mutually orthogonal (one-sided Jacobi, Hestenes) leaves
// X = V sqrt(S): the columns ARE the rows of J0, their squared norms the eigenvalues, r0_i = x_i^T b / S_i, and X X^T = A' holds after
// every rotation, so J0^T J0 = A' to rounding whatever the state of convergence (which only decides the eps threshold and r0).
//   1. X from a diagonally pivoted Cholesky without row exchanges: step k takes the largest remaining diagonal entry j, column
//      k of X = A[:, j] / sqrt(A[j][j]) over the rows not yet taken, A -= x_k x_k^T. It stops at the first pivot <= 0: a
//      semi-definite A' (the 4 gauge directions of a first prior) gives r < n columns. Jacobi then works on X^T X = L^T L, one
//      step of the LR iteration past A'.
//   2. Pair i of a step = the two columns lane i holds in registers (its "top" and "bottom"), 12 rows of them per wave (8 of the 16
//      waves). A step: every wave leaves the three partial dot products of its rows in LDS, wave 0 sums them and computes the rotation
//      of every pair, every wave rotates its rows and makes the round-robin tournament's move inside the registers (tops one lane up,
//      bottoms one lane down: v_mov_b32_dpp wave_shr:1 / wave_shl:1). The matrix never touches LDS inside the sweeps.
//   3. Sweeps end when the largest |cos| a sweep met is <= 1e-7 (quadratic convergence: the sweep itself took it to 1e-14).
// 1 / sqrt(x) to double rounding: v_rsq_f64 (about 2^-23) and two Newton steps
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * (1.5 - hx * y * y);
  y = y * (1.5 - hx * y * y);
  return y;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_step(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return fmax(v, __hiloint2double(hi, lo));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double old, double src) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// max over the wave of non-negative values (0.0 is the identity the DPP steps shift in), the same in every lane: prefix maxima within
// each row of 16 lanes (row_shr 1, 2, 4, 8), row_bcast15 / row_bcast31 across rows, lane 63 holds the result
__device__ __forceinline__ double wave_max_nonneg(double v) {
  v = dpp_max_step<0x111, 0xF>(v);
  v = dpp_max_step<0x112, 0xF>(v);
  v = dpp_max_step<0x114, 0xF>(v);
  v = dpp_max_step<0x118, 0xF>(v);
  v = dpp_max_step<0x142, 0xA>(v);
  v = dpp_max_step<0x143, 0xC>(v);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// The sweeps of prior_factor_lds for JR row slots per wave (rows wv + JW * i, i < JR; JW * JR >= n): see the comment above it.
constexpr int JW = 8;   // waves that hold rows
template <int JR>
__device__ __forceinline__ void jacobi_sweeps(double *P, double *part /* [3][JW][48] */, double *cs, int *conv_s, int ne) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, h = ne >> 1;
  const bool act = lane < h && wv < JW;
  const int lc = min(lane, 47);
  double t[JR], b[JR];   // this wave's JR rows of the two columns of pair `lane`
#pragma unroll
  for (int i = 0; i < JR; ++i) {
    const double tv = P[lc * MG_LD + wv % JW + JW * i], bv = P[min(h + lc, MG_NMAX - 1) * MG_LD + wv % JW + JW * i];
    t[i] = act ? tv : 0.0;
    b[i] = act ? bv : 0.0;
  }
  __syncthreads();
  const bool first = lane == 0, last = lane == h - 1;
  double na = 0.0, nb = 0.0;   // wave 0: squared norms of the two columns of pair `lane`
  for (int sweep = 0; sweep < 30; ++sweep) {
    double mx = 0.0;   // largest cos^2 this sweep rotated away (wave 0)
    // squared column norms from the data once per sweep; inside the sweep they follow the rotations in closed form
    // (|x_p'|^2 = c^2 a - 2 c s g + s^2 b, |x_q'|^2 = s^2 a + 2 c s g + c^2 b) and travel with their columns
    if (wv < JW) {
      double a_ = 0.0, b_ = 0.0;
#pragma unroll
      for (int i = 0; i < JR; ++i) { a_ += t[i] * t[i]; b_ += b[i] * b[i]; }
      if (act) { part[wv * 48 + lane] = a_; part[(JW + wv) * 48 + lane] = b_; }
    }
    __syncthreads();
    if (wv == 0) {
      na = 0.0; nb = 0.0;
#pragma unroll
      for (int w = 0; w < JW; ++w) { na += part[w * 48 + lc]; nb += part[(JW + w) * 48 + lc]; }
    }
    for (int step = 0; step < ne - 1; ++step) {
      if (wv < JW) {
        double g_ = 0.0;
#pragma unroll
        for (int i = 0; i < JR; ++i) g_ += t[i] * b[i];
        if (act) part[(2 * JW + wv) * 48 + lane] = g_;
      }
      __syncthreads();
      if (wv == 0) {   // all 64 lanes: the norms move by DPP like the columns; lanes >= h carry values nobody reads
        double sg = 0.0;
#pragma unroll
        for (int w = 0; w < JW; ++w) sg += part[(2 * JW + w) * 48 + lc];
        // tan 2 theta = 2 sg / (nb - na), |theta| <= pi / 4: cos 2 theta = |zeta| / hyp, c = sqrt((1 + cos 2 theta) / 2), s = sin 2 theta / (2 c)
        double c = 1.0, sn = 0.0;
        const double ab = na * nb, g2 = sg * sg;
        const double zeta = nb - na, gam = 2.0 * sg, hyp2 = zeta * zeta + gam * gam;
        if (act && g2 > 1e-28 * ab && hyp2 > 1e-290) {
          mx = fmax(mx, g2 * __builtin_amdgcn_rcp(ab));
          const double rh = rsqrt_nr(hyp2);
          const double c2 = fabs(zeta) * rh, s2 = (zeta >= 0.0 ? gam : -gam) * rh;
          const double hc = 0.5 + 0.5 * c2, rc = rsqrt_nr(hc);
          c = hc * rc;
          sn = 0.5 * s2 * rc;
        }
        if (act) { cs[2 * lane] = c; cs[2 * lane + 1] = sn; }
        const double cc = c * c, ss = sn * sn, x2 = 2.0 * c * sn * sg;
        const double an = fmax(cc * na - x2 + ss * nb, 0.0), bn = fmax(ss * na + x2 + cc * nb, 0.0);
        na = dpp_mov<0x138>(an, first ? bn : an);
        const double bd = dpp_mov<0x130>(bn, bn);
        nb = last ? an : bd;
      }
      __syncthreads();
      if (wv < JW) {
        // rotate, then the round-robin move inside the registers: tops go one lane up (lane 0 keeps its own, lane 1 takes lane 0's
        // bottom), bottoms one lane down (lane h - 1 takes its own top). All 64 lanes run this: a DPP move reads its neighbour's register.
        const double c = cs[2 * lc], sn = cs[2 * lc + 1];
#pragma unroll
        for (int i = 0; i < JR; ++i) {
          const double tn = c * t[i] - sn * b[i], bn = sn * t[i] + c * b[i];
          const double send = first ? bn : tn;
          const double bd = dpp_mov<0x130>(bn, bn);     // wave_shl:1
          b[i] = last ? tn : bd;
          t[i] = dpp_mov<0x138>(tn, send);              // wave_shr:1 (tn's last use: the move can land in its register)
        }
      }
    }
    if (wv == 0) {
      mx = wave_max_nonneg(mx);
      if (lane == 0) *conv_s = (mx <= 1e-14) ? 1 : 0;
    }
    __syncthreads();
    if (*conv_s) break;
  }
  if (act) {
#pragma unroll
    for (int i = 0; i < JR; ++i) { P[lane * MG_LD + wv + JW * i] = t[i]; P[(h + lane) * MG_LD + wv + JW * i] = b[i]; }
  }
}

// factor_form (vilo_set_prior_form(ctx, VILO_PRIOR_FACTOR): callers that never look at J0 itself — a resident prior pool): the prior is
// only ever used through J0^T J0, J0^T r0 and |r0|^2, which any X with X X^T = A' gives alike with J0 = X^T, r0 = X^-1 b — the reference's
// sqrt(S) V^T is Q X^T for an orthogonal Q. What the eigen form adds is the threshold: eigenvalues <= eps are dropped
// (marginalization_factor.cpp:297-305). So the pivoted Cholesky factor is taken as it stands IF it has full rank AND lambda_min(A') > eps is
// certain: lambda_min = 1 / |X^-1|_2^2 >= 1 / |X^-1|_F^2, and X^-1 comes out of the same forward substitutions that give r0 (n unit
// right-hand sides beside b, 8 lanes per column; ~30 k cycles instead of the sweeps' 1.5 M). Anything else goes on to the sweeps.
__device__ void prior_factor_lds(double *Ar /* n x n, ld MG_LD, bitwise symmetric, destroyed */, double *P /* MG_NMAX x MG_LD */, const double *br, int n, double eps,
                                 double *cs /* LDS [96] */, double *J0, double *r0, int *status_w, long long *clk_w /* [8] or null */, bool factor_form) {
  __shared__ int bad, conv, fast_ok;
  __shared__ int piv[MG_NMAX];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int ty = tid >> 5, tx = tid & 31;
  for (int e = tid; e < MG_NMAX * MG_LD; e += MGT) P[e] = 0.0;
  if (tid == 0) bad = 0;
  __syncthreads();
  {
    bool nf = false;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        if (ty + 32 * a < n && tx + 32 * b < n) nf |= !isfinite(Ar[(ty + 32 * a) * MG_LD + tx + 32 * b]);
    if (tid < n) nf |= !isfinite(br[tid]);
    if (nf) bad = 1;
  }
  // ---- pivoted Cholesky, left-looking, one barrier per step. A' stays as it is; the running diagonal lives in LDS (dg), the finished
  //      columns of X in LDS (P, for the pivot row) and in registers: thread (i, q) = (tid / 8, tid % 8) keeps X[i][q + 8 m], m < 12.
  //      Step k: every wave finds the pivot j for itself (diagonal entry with its index in the low 7 mantissa bits, so one DPP max decides
  //      value and index; ties go to the smaller index; taken rows carry 0), then x_i = (A[i][j] - sum_k' X[i][k'] X[j][k']) / sqrt(d_j):
  //      12 products per thread (columns >= k of P are still zero, so the sum needs no bound) and a sum over the 8 lanes of the row. ----
  int r = 0;
  {
    __shared__ double dgb[2][MG_NMAX];   // the diagonal, double-buffered: step k reads [k & 1] while the finished rows write [(k + 1) & 1]
    __shared__ double dinv0[MG_NMAX];   // 1 / (the diagonal A' started with): what a remaining diagonal entry is measured against (factor form)
    if (tid < MG_NMAX) {
      const double d_ = (tid < n) ? Ar[tid * (MG_LD + 1)] : 0.0;
      dgb[0][tid] = d_;
      dinv0[tid] = (d_ > 0.0) ? 1.0 / d_ : 0.0;
    }
    __syncthreads();
    const int ri = tid >> 3, rq = tid & 7, ric = min(ri, MG_NMAX - 1);
    bool taken = ri >= n;
    double xs[12];
#pragma unroll
    for (int m = 0; m < 12; ++m) xs[m] = 0.0;
    const int l1 = min(lane + 64, MG_NMAX - 1);
    // columns in blocks of 8 with the block index a compile-time constant: the sum runs over the blocks that exist, and the new entry
    // goes into a register the code names
    bool done = false;
    auto block = [&](auto mbc) {
      constexpr int mb = decltype(mbc)::value;
      if (done || 8 * mb >= n) return;
      for (int k = 8 * mb; k < min(8 * mb + 8, n); ++k) {
        const double *dg = dgb[k & 1];
        const double d0 = dg[lane], d1 = (lane + 64 < MG_NMAX) ? dg[l1] : 0.0;
        double best = 0.0;
        if (d0 > 0.0) best = __hiloint2double(__double2hiint(d0), (__double2loint(d0) & ~127) | (127 - lane));
        if (d1 > 0.0) best = fmax(best, __hiloint2double(__double2hiint(d1), (__double2loint(d1) & ~127) | (63 - lane)));
        best = wave_max_nonneg(best);
        if (!(best > 0.0)) { done = true; break; }
        // factor form: what is left of A' is positive semi-definite with trace <= (n - k) * (largest diagonal entry); once that is <= eps
        // every eigenvalue of the remainder is, and the eigen form would drop them too (a sequence's prior never fixes the four gauge
        // directions: their pivots come out as +- 1e-9 noise, and a factor that carried them on could not be certified below)
        if (factor_form) {
          if (best * (double)(n - k) <= eps) { done = true; break; }
          // ... or when every remaining diagonal entry is rounding noise of the entry it started as (<= 1e-12 of it: n u times the growth
          // of the elimination) — with information up to 1e15 in A' the gauge pivots come out as +- 1e-2, far above eps, and an eigen
          // decomposition keeps or drops such a direction by the sign its noise happens to have
          const double rel = wave_max_nonneg(fmax(d0 * dinv0[lane], d1 * ((lane + 64 < MG_NMAX) ? dinv0[l1] : 0.0)));
          if (rel <= 1e-12) { done = true; break; }
        }
        const int j = 127 - (__double2loint(best) & 127);
        const double pv = dg[j], aij = Ar[ric * MG_LD + j];
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m <= mb; ++m) acc += xs[m] * P[(rq + 8 * m) * MG_LD + j];   // columns >= k of P, if a faster wave wrote them, meet xs = 0
        acc += dpp_mov<0xB1>(0.0, acc);    // quad_perm [1, 0, 3, 2]
        acc += dpp_mov<0x4E>(0.0, acc);    // quad_perm [2, 3, 0, 1]
        acc += dpp_mov<0x141>(0.0, acc);   // row_half_mirror: the other quad of the 8 lanes
        const double rs = rsqrt_nr(pv);
        const double x = taken ? 0.0 : (ri == j ? pv * rs : (aij - acc) * rs);
        taken |= (ri == j);
        xs[mb] = (rq == (k & 7)) ? x : xs[mb];
        if (rq == 0 && ri < MG_NMAX) {
          P[k * MG_LD + ri] = x;
          dgb[(k + 1) & 1][ri] = taken ? 0.0 : dg[ri] - x * x;
        }
        if (tid == 0) piv[k] = j;
        __syncthreads();
        ++r;
      }
    };
    block(std::integral_constant<int, 0>{}); block(std::integral_constant<int, 1>{}); block(std::integral_constant<int, 2>{});
    block(std::integral_constant<int, 3>{}); block(std::integral_constant<int, 4>{}); block(std::integral_constant<int, 5>{});
    block(std::integral_constant<int, 6>{}); block(std::integral_constant<int, 7>{}); block(std::integral_constant<int, 8>{});
    block(std::integral_constant<int, 9>{}); block(std::integral_constant<int, 10>{}); block(std::integral_constant<int, 11>{});
  }
  __syncthreads();
  if (clk_w && tid == 0) clk_w[7] = (long long)__builtin_readcyclecounter();
  if (factor_form && r >= 1) {
    // X (n x r, r <= n: a semi-definite A' — the four gauge directions no prior of a sequence ever fixes — stops the factorisation at the
    // first pivot <= 0, and what is left of A' then is rounding noise the eigen form drops as well). Its pivot rows form a lower triangle
    // X_p in pivot order; sigma_min(X) >= sigma_min(X_p) >= 1 / |X_p^-1|_F. X_p y = v by forward substitution: step k fixes y[k] from row
    // piv[k]. Column c < r: v = e_c (a column of X_p^-1); column r: v = b on the pivot rows (b lies in the range of A' up to rounding, so
    // J0^T r0 = X X_p^-1 b_p reproduces it). Thread (c, q) = (tid / 8, tid % 8) sums the products k' = q mod 8; Y[k][c] in the spent A'.
    double *Y = Ar;
    const int c = tid >> 3, q = tid & 7, cc = min(c, r);
    double ssq = 0.0;
    for (int k = 0; k < r; ++k) {
      const int j = piv[k];
      double acc = 0.0;
      for (int kk = q; kk < k; kk += 8) acc += P[kk * MG_LD + j] * Y[kk * MG_LD + cc];
      acc += dpp_mov<0xB1>(0.0, acc);
      acc += dpp_mov<0x4E>(0.0, acc);
      acc += dpp_mov<0x141>(0.0, acc);
      if (q == 0 && c <= r) {
        const double rhs = (c < r) ? (c == k ? 1.0 : 0.0) : br[j];
        const double y = (rhs - acc) / P[k * MG_LD + j];
        Y[k * MG_LD + c] = y;
        if (c < r) ssq += y * y;
      }
      lds_fence();   // (the column's next step reads what lane q == 0 of these 8 lanes just wrote: same wave, LDS operations in order)
    }
    if (tid == 0) fast_ok = 0;
    if (tid < MG_NMAX) cs[tid] = 0.0;
    __syncthreads();
    if (q == 0 && c < r) cs[c] = ssq;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int i = 0; i < r; ++i) tot += cs[i];
      fast_ok = (isfinite(tot) && tot * eps < 1.0) ? 1 : 0;   // 1 / |X_p^-1|_F^2 > eps: no eigenvalue of X X^T in (0, eps]
    }
    __syncthreads();
    if (fast_ok) {
      if (tid < n) {
        const double rv = (tid < r) ? Y[tid * MG_LD + r] : 0.0;
        r0[tid] = rv;
        if (!isfinite(rv) || bad) *status_w = 1;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int i = ty + 32 * a;
        if (i >= n) continue;
        for (int j = tx; j < n; j += 32) J0[i * n + j] = (i < r) ? P[i * MG_LD + j] : 0.0;
      }
      if (clk_w && tid == 0) clk_w[5] = (long long)__builtin_readcyclecounter();
      return;
    }
    __syncthreads();
  }
  const int ne = max(4, (r + 1) & ~1), h = ne >> 1;   // zero columns fill up; at least two pairs, so the move below has no special case
  static_assert(JW * 12 >= MG_NMAX && JW <= MGT / 64, "row split of the Jacobi sweeps");
  double *part = Ar;   // [3][JW][48] partial dot products; A' is spent
  if (n <= JW * 10) jacobi_sweeps<10>(P, part, cs, &conv, ne);
  else if (n <= JW * 11) jacobi_sweeps<11>(P, part, cs, &conv, ne);
  else jacobi_sweeps<12>(P, part, cs, &conv, ne);
  __syncthreads();
  if (clk_w && tid == 0) clk_w[5] = (long long)__builtin_readcyclecounter();
  // eigenvalue of position i = |x_i|^2; after whole sweeps every column is back at the position it started from, any order of the rows of J0 is as good
  if (tid < MG_NMAX) {
    double S = 0.0, xb = 0.0;
    if (tid < ne)
      for (int row = 0; row < n; ++row) { const double v = P[tid * MG_LD + row]; S += v * v; xb += v * br[row]; }
    cs[tid] = S;
    if (tid < n) {
      const double rv = (S > eps) ? xb / S : 0.0;
      r0[tid] = rv;
      if (!isfinite(rv) || bad) *status_w = 1;
    }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int i = ty + 32 * a;
    if (i >= n) continue;
    const bool keep = cs[i] > eps;
    for (int j = tx; j < n; j += 32) J0[i * n + j] = keep ? P[i * MG_LD + j] : 0.0;
  }
}

__global__ void __launch_bounds__(MGT) k_marginalize_lds(BatchDev bd, const MargWin *mw, const int *drop_lm, int max_l0, double *J0_out,
                                                         double *r0_out, int *status, int *need_general, long long *clk /* [W][8] or null */, int factor_form) {
  extern __shared__ double ml[];
  double *A1 = ml, *tile = ml + MG_TILE_OFF;
  double *b1 = ml + MG_R0, *dinv = b1 + MG_TMAX, *deps = dinv + MG_TILE, *gl = deps + MG_TILE, *Ce = gl + MG_TILE, *cs = Ce + 19 * 19;
  double *yb = cs + 2 * 48, *br = yb + 19;
  __shared__ int act_a[80], act_t[80];
  __shared__ int n_act, fail;
  __shared__ double dx[VILO_MAX_PRIOR_DIM];
  const int win = blockIdx.x, tid = threadIdx.x;
  const MargWin &M = mw[win];
  const WinMeta wm = bd.win[win];
  const int m = M.m, n = M.n, L0 = M.n_drop_lm, md = m - L0, T = md + n;
  if (m == 0 || n == 0) return;
  if (T > MG_TMAX || n > MG_NMAX || md > 19) { if (tid == 0) need_general[win] = 1; return; }
  const double eps = 1e-8;
  const double *x = bd.x + (size_t)win * XSTRIDE;
  auto stamp = [&](int i) { if (clk && tid == 0) clk[win * 8 + i] = (long long)__builtin_readcyclecounter(); };
  stamp(0);
  // index of a camera dim in [dense dropped | kept], -1 if absent
  auto loc = [&](int ci) { const int i = M.cdmap[ci]; return i < 0 ? -1 : (i < md ? i : i - L0); };
  for (int e = tid; e < T * T; e += MGT) A1[e] = 0.0;
  for (int e = tid; e < T; e += MGT) b1[e] = 0.0;
  if (tid == 0) { n_act = 0; fail = 0; }
  __syncthreads();
  auto add = [&](int ci, int cj, double v) {
    const int i = loc(ci), j = loc(cj);
    if (i >= 0 && j >= 0) A1[i * T + j] += v;
  };
  // ---- prior factor (marginalization_factor.cpp:347-395 evaluated at the current state): J^T J = H, J^T r = b0 + H dx ----
  if (M.prior_n > 0) {
    const int pn = M.prior_n;
    const double *Hp = bd.prior_H + (size_t)win * 96 * 96, *b0 = bd.prior_b0 + (size_t)win * 96;
    const int *pmap = bd.prior_map + (size_t)win * 96;
    if (tid < wm.prior_nb)
      prior_dx(x + bd.prior_bstate[win * 40 + tid], bd.prior_x0 + (size_t)win * 280 + bd.prior_bxoff[win * 40 + tid],
               bd.prior_bsize[win * 40 + tid], dx + bd.prior_bidx[win * 40 + tid]);
    __syncthreads();
    for (int e = tid; e < pn * pn; e += MGT) add(pmap[e / pn], pmap[e % pn], Hp[e]);
    for (int i = tid; i < pn; i += MGT) {
      double sacc = b0[i];
      for (int q = 0; q < pn; ++q) sacc += Hp[(size_t)q * pn + i] * dx[q];
      const int t = loc(pmap[i]);
      if (t >= 0) b1[t] += sacc;
    }
    __syncthreads();
  }
  if (M.mode == 0) {
    // ---- IMULegFactor / IMUFactor between frames 0 and 1: whitened [J | r] staged through LDS ----
    if (M.has_imu) {
      const double *lin = bd.imu_lin + (size_t)win * 10 * 31 * 39;
      for (int e = tid; e < 31 * 39; e += MGT) tile[e] = lin[e];
      __syncthreads();
      for (int e = tid; e < 39 * 39; e += MGT) {
        const int a = e / 39, c = e % 39;
        if (a == 38) continue;
        double sacc = 0.0;
        for (int i = 0; i < 31; ++i) sacc += tile[i * 39 + a] * tile[i * 39 + c];
        auto cdof = [](int cc) { return cc < 6 ? cc : (cc < 19 ? CD_B0 + (cc - 6) : (cc < 25 ? 6 + (cc - 19) : CD_B0 + 13 + (cc - 25))); };
        if (c == 38) { const int t = loc(cdof(a)); if (t >= 0) b1[t] += sacc; }
        else add(cdof(a), cdof(c), sacc);
      }
      __syncthreads();
    }
    // ---- visual factors of the landmarks that start in frame 0: camera-side Gram slots of the s = 0 chunks ----
    for (int ch = 0; ch < wm.n_chunks; ++ch) {
      const ChunkMeta cm = bd.chunk[wm.chunk_off + ch];
      if (cm.s != 0) continue;
      for (int t = 0; t < cm.kmax; ++t) {
        const double *gs = bd.gram + (size_t)(cm.gram_off + t) * VILO_GRAM;
        for (int e = tid; e < VILO_GRAM26; e += MGT) {   // the 26-column view of the slot
          int a = 0, rem = e;
          while (rem >= 26 - a) { rem -= 26 - a; ++a; }
          const int bc = a + rem;
          if (t == 0 && ((a >= 6 && a < 12) || (bc >= 6 && bc < 12))) continue;
          auto cdof = [&](int c) { return c < 6 ? c : (c < 12 ? 6 * t + (c - 6) : (c < 18 ? CD_EX0 + c - 12 : (c < 24 ? CD_EX1 + c - 18 : CD_TD))); };
          double sg;
          const int ge = gram26_index(a, bc, sg);
          const double v = sg * gs[ge];
          if (bc == 25) { if (a < 25) { const int q = loc(cdof(a)); if (q >= 0) b1[q] += v; } }
          else {
            add(cdof(a), cdof(bc), v);
            if (cdof(a) != cdof(bc)) add(cdof(bc), cdof(a), v);
          }
        }
        __syncthreads();
      }
    }
  }
  stamp(1);
  // ---- certificate copy: C - eps I (dense dropped block before the landmarks are folded in) ----
  for (int e = tid; e < md * md; e += MGT) Ce[e] = A1[(e / md) * T + e % md] - ((e / md == e % md) ? eps : 0.0);
  // camera dims that exist in this problem (rows of the landmark coupling W)
  if (tid == 0) {
    int k = 0;
    for (int a = 0; a < 79; ++a) { const int t = loc(a); if (t >= 0) { act_a[k] = a; act_t[k] = t; ++k; } }
    n_act = k;
  }
  __syncthreads();
  // ---- eliminate the dropped landmarks: A1 -= W D^-1 W^T, b1 -= W D^-1 g, tile by tile ----
  if (L0 > 0) {
    const int na = n_act;
    const double *wl = bd.lm_w + 80 * (size_t)wm.lm_off;
    for (int l0 = 0; l0 < L0; l0 += MG_TILE) {
      const int nl = min(MG_TILE, L0 - l0);
      if (tid < nl) {
        const int l = drop_lm[(size_t)win * max_l0 + l0 + tid];
        const double D = bd.lm_E[wm.lm_off + l];
        if (!(D - eps > 0.0)) fail = 1;
        dinv[tid] = 1.0 / D; deps[tid] = 1.0 / (D - eps); gl[tid] = bd.lm_gbuf[0][wm.lm_off + l];
      }
      for (int e = tid; e < na * MG_TILE; e += MGT) {
        const int k = e / MG_TILE, j = e % MG_TILE;
        tile[e] = (j < nl) ? wl[(size_t)act_a[k] * wm.L + drop_lm[(size_t)win * max_l0 + l0 + j]] : 0.0;
      }
      __syncthreads();
      // W D^-1 W^T of the tile on the FP64 matrix cores: one 16 x 16 block of the (compact) active-dimension index per wave, K = the
      // 32 landmarks of the tile (8 k-steps); lane (lr, lk) supplies W[16 K1 + lr][4 kk + lk] / D and W[16 K2 + lr][4 kk + lk]
      {
        const int wv_ = tid >> 6, lane_ = tid & 63, lr = lane_ & 15, lk = lane_ >> 4;
        const int nb = (na + 15) >> 4;
        for (int blk = wv_; blk < (nb * (nb + 1)) / 2; blk += MGT / 64) {
          int K1 = 0;
          while (((K1 + 1) * (K1 + 2)) / 2 <= blk) ++K1;
          const int K2 = blk - (K1 * (K1 + 1)) / 2;
          const int ra = 16 * K1 + lr, rb = 16 * K2 + lr;
          mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kk = 0; kk < MG_TILE / 4; ++kk) {
            const int j = 4 * kk + lk;
            const double wa = (ra < na && j < nl) ? tile[ra * MG_TILE + j] * dinv[j] : 0.0;
            const double wb = (rb < na && j < nl) ? tile[rb * MG_TILE + j] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, wb, acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k1 = 16 * K1 + lk + 4 * r, k2 = 16 * K2 + lr;
            if (k1 < na && k2 < na && (K1 != K2 || k2 <= k1)) {
              const int t1 = act_t[k1], t2 = act_t[k2];
              A1[t1 * T + t2] -= acc[r];
              if (t1 != t2) A1[t2 * T + t1] -= acc[r];
            }
          }
        }
      }
      // the certificate's copy of the dense dropped block takes the same update with 1 / (D - eps)
      for (int e = tid; e < na * (na + 1) / 2; e += MGT) {
        int k1 = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while ((k1 + 1) * (k1 + 2) / 2 <= e) ++k1;
        while (k1 * (k1 + 1) / 2 > e) --k1;
        const int k2 = e - k1 * (k1 + 1) / 2;
        const int t1 = act_t[k1], t2 = act_t[k2];
        if (!(t1 < md && t2 < md)) continue;
        double se = 0.0;
        for (int j = 0; j < nl; ++j) se += tile[k1 * MG_TILE + j] * tile[k2 * MG_TILE + j] * deps[j];
        Ce[t1 * md + t2] -= se;
        if (t1 != t2) Ce[t2 * md + t1] -= se;
      }
      if (tid < na) {
        double sacc = 0.0;
        for (int j = 0; j < nl; ++j) sacc += tile[tid * MG_TILE + j] * gl[j] * dinv[j];
        b1[act_t[tid]] -= sacc;
      }
      __syncthreads();
    }
  }
  stamp(2);
  // ---- certificate: Amm - eps I positive definite <=> Cholesky of its landmark-reduced dense block has positive pivots ----
  chol_lds(Ce, md, md, &fail);
  if (fail) { if (tid == 0) need_general[win] = 1; return; }
  // ---- eliminate the dense dropped dims: A' = Arr - Arm C^-1 Amr with C = L L^T ----
  chol_lds(A1, T, md, &fail);
  if (fail) { if (tid == 0) need_general[win] = 1; return; }
  stamp(3);
  double *Y = tile;   // md x n, Y = L^-1 A1[0:md, md:T]
  if (tid <= n) {
    for (int i = 0; i < md; ++i) {
      double sacc = (tid < n) ? 0.5 * (A1[i * T + md + tid] + A1[(md + tid) * T + i]) : b1[i];
      for (int k = 0; k < i; ++k) sacc -= A1[i * T + k] * ((tid < n) ? Y[k * n + tid] : yb[k]);
      sacc /= A1[i * T + i];
      if (tid < n) Y[i * n + tid] = sacc; else yb[i] = sacc;
    }
  }
  __syncthreads();
  // A' = Arr - Y^T Y (lower triangle, mirrored: SelfAdjointEigenSolver reads the lower triangle) on the FP64 matrix cores: 16 x 16 blocks
  // of the kept dimensions, K = the <= 19 dense dropped dims (5 k-steps); the blocks stay in registers until every wave has read its part
  // of A1, then go to the compact layout (ld = MG_LD) that overlaps it
  constexpr int NBN = (MG_NMAX + 15) / 16, NBLK = (NBN * (NBN + 1)) / 2, BPW = (NBLK + MGT / 64 - 1) / (MGT / 64);
  mfma_d4 keep[BPW];
  {
    const int wv_ = tid >> 6, lane_ = tid & 63, lr = lane_ & 15, lk = lane_ >> 4;
    const int nbn = (n + 15) >> 4;
#pragma unroll
    for (int u = 0; u < BPW; ++u) {
      const int blk = wv_ + u * (MGT / 64);
      keep[u] = mfma_d4{0.0, 0.0, 0.0, 0.0};
      if (blk >= (nbn * (nbn + 1)) / 2) continue;
      int I = 0;
      while (((I + 1) * (I + 2)) / 2 <= blk) ++I;
      const int J = blk - (I * (I + 1)) / 2;
      mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 5; ++kk) {
        const int k = 4 * kk + lk;
        const double ya = (k < md && 16 * I + lr < n) ? Y[k * n + 16 * I + lr] : 0.0;
        const double yc = (k < md && 16 * J + lr < n) ? Y[k * n + 16 * J + lr] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ya, yc, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + lk + 4 * r, j = 16 * J + lr;
        keep[u][r] = (i < n && j < n && j <= i) ? A1[(md + i) * T + md + j] - acc[r] : 0.0;
      }
    }
  }
  if (tid < n) {
    double sacc = b1[md + tid];
    for (int k = 0; k < md; ++k) sacc -= Y[k * n + tid] * yb[k];
    br[tid] = sacc;
  }
  __syncthreads();
  double *Ar = ml, *V2 = ml + MG_NMAX * MG_LD;
  {
    const int wv_ = tid >> 6, lane_ = tid & 63, lr = lane_ & 15, lk = lane_ >> 4;
    const int nbn = (n + 15) >> 4;
#pragma unroll
    for (int u = 0; u < BPW; ++u) {
      const int blk = wv_ + u * (MGT / 64);
      if (blk >= (nbn * (nbn + 1)) / 2) continue;
      int I = 0;
      while (((I + 1) * (I + 2)) / 2 <= blk) ++I;
      const int J = blk - (I * (I + 1)) / 2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + lk + 4 * r, j = 16 * J + lr;
        if (i < n && j < n && j <= i) { Ar[i * MG_LD + j] = keep[u][r]; Ar[j * MG_LD + i] = keep[u][r]; }
      }
    }
  }
  __syncthreads();
  stamp(4);
  double *J0 = J0_out + (size_t)win * VILO_MAX_PRIOR_DIM * VILO_MAX_PRIOR_DIM, *r0 = r0_out + (size_t)win * VILO_MAX_PRIOR_DIM;
  prior_factor_lds(Ar, V2, br, n, eps, cs, J0, r0, &status[win], clk ? clk + win * 8 : nullptr, factor_form != 0);
  stamp(6);
}

// flag[w] = the IMU factor of interval 0 is live and its covariance had no sqrt_info
__global__ void k_marg_imu0_bad(int W, const int *prep_bad, const unsigned char *imu_skip, int *flag) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < W) flag[w] = (!imu_skip[(size_t)w * 10] && prep_bad[(size_t)w * 10]) ? 1 : 0;
}

// new priors of pooled windows: device to device into their slot (n x n packed, ld n); src >= 0: the unchanged prior moves
// from slot src to slot dst (MARGIN_SECOND_NEW with nothing to drop)
__global__ void __launch_bounds__(256) k_prior_scatter(int W, const MargWin *mw, const int *dst, const int *src, const double *J0, const double *r0, double *pJ,
                                                       double *pr) {
  const int w = blockIdx.x;
  if (w >= W || dst[w] < 0) return;
  double *dj = pJ + (size_t)dst[w] * 96 * 96, *dr = pr + (size_t)dst[w] * 96;
  if (src[w] >= 0) {
    const double *sj = pJ + (size_t)src[w] * 96 * 96, *sr = pr + (size_t)src[w] * 96;
    for (int e = threadIdx.x; e < 96 * 96; e += 256) dj[e] = sj[e];
    for (int e = threadIdx.x; e < 96; e += 256) dr[e] = sr[e];
    return;
  }
  const int n = mw[w].n;
  const double *sj = J0 + (size_t)w * VILO_MAX_PRIOR_DIM * VILO_MAX_PRIOR_DIM, *sr = r0 + (size_t)w * VILO_MAX_PRIOR_DIM;
  for (int e = threadIdx.x; e < n * n; e += 256) dj[e] = sj[e];
  for (int e = threadIdx.x; e < n; e += 256) dr[e] = sr[e];
}

__device__ v3 R2ypr_deg(const m3 &R) {
  const v3 nn = mk3(R.a[0], R.a[3], R.a[6]), o = mk3(R.a[1], R.a[4], R.a[7]), a = mk3(R.a[2], R.a[5], R.a[8]);
  const double y = atan2(nn.y, nn.x);
  const double p = atan2(-nn.z, nn.x * cos(y) + nn.y * sin(y));
  const double r = atan2(a.x * sin(y) - a.y * cos(y), -o.x * sin(y) + o.y * cos(y));
  return mk3(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);   // `ypr / M_PI * 180.0` as utility.h:98 writes it (an ulp from `* (180 / pi)`)
}
__device__ quat quat_from_R_dev(const m3 &m) {
  quat q;
  double t = m.a[0] + m.a[4] + m.a[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m.a[7] - m.a[5]) * t; q.y = (m.a[2] - m.a[6]) * t; q.z = (m.a[3] - m.a[1]) * t;
  } else {
    int i = 0;
    if (m.a[4] > m.a[0]) i = 1;
    if (m.a[8] > m.a[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m.a[4 * i] - m.a[4 * j] - m.a[4 * k] + 1.0);
    double vi = 0.5 * t;
    t = 0.5 / t;
    q.w = (m.a[3 * k + j] - m.a[3 * j + k]) * t;
    const double vj = (m.a[3 * j + i] + m.a[3 * i + j]) * t, vk = (m.a[3 * k + i] + m.a[3 * i + k]) * t;
    q.x = (i == 0) ? vi : (j == 0 ? vj : vk);
    q.y = (i == 1) ? vi : (j == 1 ? vj : vk);
    q.z = (i == 2) ? vi : (j == 2 ? vj : vk);
  }
  return q;
}

__device__ void k_gauge_fix_body(int F, const double *bp /*[7]*/, double *pw /*[F][7]*/, double *sw /*[F][9]*/, double *exw /*[2][7]*/) {
  const int t = threadIdx.x;
  const m3 Rs0 = qR(ldq_pose(bp));
  const m3 R00 = qR(ldq_pose(pw));
  const v3 o0 = R2ypr_deg(Rs0), o00 = R2ypr_deg(R00);
  const double yd = (o0.x - o00.x) / 180.0 * M_PI;
  m3 rot = m3_eye();
  rot.a[0] = cos(yd); rot.a[1] = -sin(yd); rot.a[3] = sin(yd); rot.a[4] = cos(yd);
  if (fabs(fabs(o0.y) - 90) < 1.0 || fabs(fabs(o00.y) - 90) < 1.0) rot = Rs0 * tr(R00);
  const v3 P0 = ld3(pw), origin = ld3(bp);
  __syncthreads();
  if (t < F) {
    double *pp = pw + 7 * t, *ss = sw + 9 * t;
    const m3 Ri = rot * qR(qnormalized(ldq_pose(pp)));
    const v3 Pi = rot * (ld3(pp) - P0) + origin;
    const v3 Vi = rot * ld3(ss);
    const quat q = quat_from_R_dev(Ri);
    __syncthreads();
    st3(pp, Pi);
    pp[3] = q.x; pp[4] = q.y; pp[5] = q.z; pp[6] = q.w;
    st3(ss, Vi);
  } else {
    __syncthreads();
    if (t < F + 2) {
      double *pp = exw + 7 * (t - F);
      const quat q = quat_from_R_dev(qR(qnormalized(ldq_pose(pp))));
      pp[3] = q.x; pp[4] = q.y; pp[5] = q.z; pp[6] = q.w;
    }
  }
}

__global__ void k_gauge_fix(int W, int F, const double *before_pose0 /*[W][7]*/, double *pose /*[W][F][7]*/, double *sb /*[W][F][9]*/,
                            double *ex /*[W][2][7]*/) {
  const int w = blockIdx.x;
  if (w >= W) return;
  k_gauge_fix_body(F, before_pose0 + 7 * w, pose + (size_t)w * F * 7, sb + (size_t)w * F * 9, ex + (size_t)w * 14);
}

// the same on the state block of a device-resident batch (pose [F][7] at XO_POSE, speed/bias [F][9] at XO_SB, extrinsics at XO_EX);
// Rs[0] / Ps[0] of before the solve come from the batch's uploaded initial states
__global__ void k_gauge_fix_x(BatchDev b) {
  const int w = blockIdx.x;
  double *x = b.x + (size_t)w * XSTRIDE;
  k_gauge_fix_body(b.win[w].n_frames, b.x0 + (size_t)w * XSTRIDE + XO_POSE, x + XO_POSE, x + XO_SB, x + XO_EX);
}

}  // namespace

int vilo_gauge_fix_batch(vilo_ctx *ctx, BatchDev &bd) {
  hipLaunchKernelGGL(k_gauge_fix_x, dim3(bd.W), dim3(64), 0, ctx->stream, bd);
  VILO_HIP(hipGetLastError());
  return VILO_OK;
}

extern "C" double vilo_last_marginalize_ms(const vilo_ctx *ctx) { return ctx ? ctx->last_marg_ms : 0.0; }
extern "C" int vilo_debug_marg_general_count(const vilo_ctx *ctx) { return ctx ? ctx->marg_general_count : 0; }

extern "C" int vilo_gauge_fix(vilo_ctx *ctx, int W, const vilo_window_state *before, vilo_window_state *after, int F) {
  if (!ctx || W <= 0 || !before || !after || F < 1 || F > VILO_MAX_FRAMES) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  std::vector<double> bp((size_t)W * 7), pose((size_t)W * F * 7), sb((size_t)W * F * 9), ex((size_t)W * 14);
  for (int w = 0; w < W; ++w) {
    memcpy(&bp[(size_t)w * 7], before[w].pose, 7 * sizeof(double));
    memcpy(&pose[(size_t)w * F * 7], after[w].pose, sizeof(double) * F * 7);
    memcpy(&sb[(size_t)w * F * 9], after[w].speed_bias, sizeof(double) * F * 9);
    memcpy(&ex[(size_t)w * 14], after[w].ex_pose, sizeof(double) * 14);
  }
  DevBuf d_bp, d_pose, d_sb, d_ex;
  VILO_HIP(d_bp.alloc(bp.size() * 8)); VILO_HIP(d_pose.alloc(pose.size() * 8)); VILO_HIP(d_sb.alloc(sb.size() * 8)); VILO_HIP(d_ex.alloc(ex.size() * 8));
  VILO_HIP(hipMemcpyAsync(d_bp.p, bp.data(), bp.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  VILO_HIP(hipMemcpyAsync(d_pose.p, pose.data(), pose.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  VILO_HIP(hipMemcpyAsync(d_sb.p, sb.data(), sb.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  VILO_HIP(hipMemcpyAsync(d_ex.p, ex.data(), ex.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_gauge_fix, dim3(W), dim3(64), 0, ctx->stream, W, F, d_bp.as<double>(), d_pose.as<double>(), d_sb.as<double>(), d_ex.as<double>());
  VILO_HIP(hipGetLastError());
  VILO_HIP(hipMemcpyAsync(pose.data(), d_pose.p, pose.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  VILO_HIP(hipMemcpyAsync(sb.data(), d_sb.p, sb.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  VILO_HIP(hipMemcpyAsync(ex.data(), d_ex.p, ex.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  for (int w = 0; w < W; ++w) {
    memcpy(after[w].pose, &pose[(size_t)w * F * 7], sizeof(double) * F * 7);
    memcpy(after[w].speed_bias, &sb[(size_t)w * F * 9], sizeof(double) * F * 9);
    memcpy(after[w].ex_pose, &ex[(size_t)w * 14], sizeof(double) * 14);
  }
  return VILO_OK;
}

// Batch internals needed here (defined in vilo_batch.hip)
struct vilo_batch;
BatchDev *vilo_batch_dev(vilo_batch *bt);
const int *vilo_batch_perm(vilo_batch *bt, int win, int *L);
int vilo_batch_scratch(vilo_ctx *ctx, vilo_batch *bt, void **p, size_t bytes);   // vilo_batch.hip: arena scratch, freed with the batch
struct View { void *p; template <class T> T *as() { return (T *)p; } };   // a typed look at part of a buffer
// per-call buffers out of the batch's arena (they go back to the context's pool with the batch)
struct ArenaBuf {
  vilo_ctx *c; vilo_batch *b; void *p = nullptr;
  hipError_t alloc(size_t n) { return vilo_batch_scratch(c, b, &p, n) == VILO_OK ? hipSuccess : hipErrorOutOfMemory; }
  template <class T> T *as() { return (T *)p; }
};

// Marginalisation of every window of an existing batch at its current device state (b.x, b.lam). `state` holds the same
// values on the host (they become keep_block_data of the new prior); modes[w]: 0 MARGIN_OLD, 1 MARGIN_SECOND_NEW, < 0 skip
// (out[w] untouched). The batch is left alive.
static int marginalize_batch(vilo_ctx *ctx, vilo_batch *bt, int W, const vilo_window_desc *in, const vilo_resident_refs *refs, const vilo_window_state *state,
                             const int *modes, vilo_prior *out) {
  int rc = VILO_OK;
  BatchDev &bd = *vilo_batch_dev(bt);
  std::vector<MargWin> mws(W);
  std::vector<std::vector<int>> kept_ids(W), kept_cd(W), kept_gs(W), kept_soff(W);
  int max_l0 = 1;
  std::vector<std::vector<int>> drops(W);
  std::vector<char> keep_prior(W, 0), skip(W, 0);
  size_t scratch_total = 0;
  for (int w = 0; w < W; ++w) {
    const vilo_window_desc &d = in[w];
    MargWin &M = mws[w];
    memset(&M, 0, sizeof(M));
    for (int i = 0; i < CD_N; ++i) M.cdmap[i] = -1;
    const int mode = modes[w];
    if (mode < 0) { M.m = 0; M.n = 0; M.mode = 0; skip[w] = 1; continue; }
    M.mode = mode;
    const int F = d.n_frames, WS = F - 1;
    const vilo_resident_refs *rf = refs ? refs + w : nullptr;
    const vilo_prior *prw = vilo_win_prior(d, rf);
    const bool has_prior = prw && prw->valid && prw->n > 0;
    M.prior_n = has_prior ? prw->n : 0;
    // which camera blocks take part (id = kind*16 + index), in the oracle's canonical order
    std::vector<int> present;   // block ids
    auto mark = [&](int id) { if (std::find(present.begin(), present.end(), id) == present.end()) present.push_back(id); };
    if (has_prior)
      for (int k = 0; k < prw->n_blocks; ++k) mark(prw->block_id[k]);
    std::vector<int> dropped_ids;
    if (mode == 0) {
      const int nkind = d.use_leg ? 3 : 2;   // pose, speed/bias (, leg bias)
      M.has_imu = vilo_win_sum_dt(d, rf, 0) < 10.0 ? 1 : 0;
      if (M.has_imu)
        for (int kind = 0; kind < nkind; ++kind) { mark(kind * 16 + 0); mark(kind * 16 + 1); }
      int L; const int *perm = vilo_batch_perm(bt, w, &L);
      for (int i = 0; i < L; ++i) {
        const int l = perm[i];
        if (d.lm_start_frame[l] != 0) continue;
        const int K = d.lm_obs_offset[l + 1] - d.lm_obs_offset[l];
        mark(VILO_BLK_POSE * 16 + 0);
        for (int t = 1; t < K; ++t) mark(VILO_BLK_POSE * 16 + t);
        mark(VILO_BLK_EX * 16 + 0); mark(VILO_BLK_EX * 16 + 1); mark(VILO_BLK_TD * 16);
      }
      // dropped landmark list in ORIGINAL feature order (oracle: ascending parameter index)
      std::vector<std::pair<int, int>> dl;
      for (int i = 0; i < L; ++i)
        if (d.lm_start_frame[perm[i]] == 0) dl.push_back({perm[i], i});
      std::sort(dl.begin(), dl.end());
      for (auto &pr : dl) drops[w].push_back(pr.second);
      M.n_drop_lm = (int)drops[w].size();
      max_l0 = std::max(max_l0, M.n_drop_lm);
      for (int kind = 0; kind < 3; ++kind)
        if (std::find(present.begin(), present.end(), kind * 16) != present.end()) dropped_ids.push_back(kind * 16);
    } else {
      if (!has_prior || std::find(present.begin(), present.end(), VILO_BLK_POSE * 16 + (WS - 1)) == present.end()) {
        // estimator.cpp:1379-1380: nothing is marginalised and last_marginalization_info stays as it is
        keep_prior[w] = has_prior ? 1 : 0;
        M.m = 0; M.n = 0;
        continue;
      }
      dropped_ids.push_back(VILO_BLK_POSE * 16 + (WS - 1));
    }
    auto blk = [&](int id, int &cd, int &ls, int &gs, int &soff) {
      const int kind = id / 16, index = id % 16;
      if (kind == VILO_BLK_POSE) { cd = 6 * index; ls = 6; gs = 7; soff = XO_POSE + 7 * index; }
      else if (kind == VILO_BLK_SB) { cd = CD_B0 + 13 * index; ls = 9; gs = 9; soff = XO_SB + 9 * index; }
      else if (kind == VILO_BLK_LB) { cd = CD_B0 + 13 * index + 9; ls = 4; gs = 4; soff = XO_LB + 4 * index; }
      else if (kind == VILO_BLK_EX) { cd = CD_EX0 + 6 * index; ls = 6; gs = 7; soff = XO_EX + 7 * index; }
      else { cd = CD_TD; ls = 1; gs = 1; soff = XO_TD; }
    };
    int pos = 0;
    std::sort(dropped_ids.begin(), dropped_ids.end());
    for (int id : dropped_ids) {
      int cd, ls, gs, soff; blk(id, cd, ls, gs, soff);
      for (int c = 0; c < ls; ++c) M.cdmap[cd + c] = pos + c;
      pos += ls;
    }
    // (without an IMU factor on interval (0, 1) — sum_dt > 10 s — and without a prior on them, speed-bias / leg-bias of frame 0 are in no
    // residual block and are not marginalised: the dense dropped part is the pose alone, as in MarginalizationInfo::addResidualBlockInfo)
    pos += M.n_drop_lm;
    M.m = pos;
    std::vector<int> kept;
    for (int id : present)
      if (std::find(dropped_ids.begin(), dropped_ids.end(), id) == dropped_ids.end()) kept.push_back(id);
    std::sort(kept.begin(), kept.end());
    for (int id : kept) {
      int cd, ls, gs, soff; blk(id, cd, ls, gs, soff);
      for (int c = 0; c < ls; ++c) M.cdmap[cd + c] = pos + c;
      kept_ids[w].push_back(id); kept_cd[w].push_back(pos - M.m); kept_gs[w].push_back(gs); kept_soff[w].push_back(soff);
      pos += ls;
    }
    M.n = pos - M.m;
    if (M.n > VILO_MAX_PRIOR_DIM || (int)kept.size() > VILO_MAX_PRIOR_BLOCKS || M.m > 1200) { return VILO_ERR_UNSUPPORTED; }
  }
  std::vector<int> drop_flat((size_t)W * max_l0, 0);
  for (int w = 0; w < W; ++w)
    for (size_t i = 0; i < drops[w].size(); ++i) drop_flat[(size_t)w * max_l0 + i] = drops[w][i];
  // One arena block for what goes up (window tables, dropped-landmark lists) and the flags that come back (status, need_general, "the
  // IMU factor of interval 0 stands on a covariance without sqrt_info"): one upload, one download per call — a one-window call per image
  // used to spend more time in its dozen synchronous copies and allocations than in its kernels.
  ArenaBuf d_blob{ctx, bt}, d_J0{ctx, bt}, d_r0{ctx, bt}, d_clk{ctx, bt};
  DevBuf d_scr;
  const bool want_clk = getenv("VILO_MARG_CLOCKS") != nullptr;
  if (want_clk && d_clk.alloc(sizeof(long long) * 8 * W) != hipSuccess) { return VILO_ERR_HIP; }
  auto fail = [&](int code) { return code; };
  const size_t off_drop = (sizeof(MargWin) * (size_t)W + 15) & ~(size_t)15, off_flags = (off_drop + sizeof(int) * drop_flat.size() + 15) & ~(size_t)15;
  const size_t blob_bytes = off_flags + sizeof(int) * 3 * (size_t)W;
  // (the context's reusable page-locked staging: it outlives the asynchronous copy on every return path)
  char *hblob = (char *)vilo_host_stage(ctx, 6, blob_bytes);
  if (!hblob) return fail(VILO_ERR_HIP);
  memset(hblob, 0, blob_bytes);
  memcpy(hblob, mws.data(), sizeof(MargWin) * (size_t)W);
  if (!drop_flat.empty()) memcpy(hblob + off_drop, drop_flat.data(), sizeof(int) * drop_flat.size());
  if (d_blob.alloc(blob_bytes) != hipSuccess || d_J0.alloc(sizeof(double) * (size_t)W * VILO_MAX_PRIOR_DIM * VILO_MAX_PRIOR_DIM) != hipSuccess ||
      d_r0.alloc(sizeof(double) * (size_t)W * VILO_MAX_PRIOR_DIM) != hipSuccess)
    return fail(VILO_ERR_HIP);
  View d_mw{d_blob.p}, d_drop{(char *)d_blob.p + off_drop}, d_status{(char *)d_blob.p + off_flags}, d_general{(char *)d_blob.p + off_flags + sizeof(int) * (size_t)W},
      d_pbad{(char *)d_blob.p + off_flags + 2 * sizeof(int) * (size_t)W};
  if (hipMemcpyAsync(d_blob.p, hblob, blob_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(VILO_ERR_HIP);
  // preMarginalize: evaluate the factors at the current state (marginalization_factor.cpp:119-138)
  (void)hipEventRecord(ctx->ev0, ctx->stream);
  rc = vilo_marg_linearize(ctx, bd);
  if (rc != VILO_OK) return fail(rc);
  const size_t lds_bytes = (size_t)MG_LDS_DOUBLES * sizeof(double);
  if (!ctx->marg_attr_set) {   // (per context = per device and host thread)
    if (hipFuncSetAttribute((const void *)k_marginalize_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return fail(VILO_ERR_HIP);
    ctx->marg_attr_set = true;
  }
  const bool force_general = getenv("VILO_MARG_GENERAL") != nullptr;   // test hook: every window through the global-memory eigen path
  std::vector<int> general(W, force_general ? 1 : 0), flags(3 * (size_t)W, 0);   // flags: status | need_general | IMU factor 0 without sqrt_info
  bool have_flags = false;
  if (!force_general) {
    have_flags = true;
    hipLaunchKernelGGL(k_marginalize_lds, dim3(W), dim3(MGT), lds_bytes, ctx->stream, bd, d_mw.as<MargWin>(), d_drop.as<int>(), max_l0, d_J0.as<double>(),
                       d_r0.as<double>(), d_status.as<int>(), d_general.as<int>(), want_clk ? d_clk.as<long long>() : nullptr, ctx->prior_form);
    if (bd.prep_bad) hipLaunchKernelGGL(k_marg_imu0_bad, dim3((W + 255) / 256), dim3(256), 0, ctx->stream, W, bd.prep_bad, bd.imu_skip, d_pbad.as<int>());
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "k_marginalize_lds launch failed"; return fail(VILO_ERR_HIP); }
    if (hipMemcpy(flags.data(), d_status.p, sizeof(int) * 3 * (size_t)W, hipMemcpyDeviceToHost) != hipSuccess) return fail(VILO_ERR_HIP);
    for (int w = 0; w < W; ++w) general[w] = flags[(size_t)W + w];
    if (want_clk) {
      long long c[8];
      if (hipMemcpy(c, d_clk.p, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess)
        fprintf(stderr, "[k_marginalize_lds] window 0 cycles: assemble %lld, landmarks %lld, cholesky %lld, schur %lld, pivoted cholesky %lld, jacobi %lld, output %lld\n", c[1] - c[0],
                c[2] - c[1], c[3] - c[2], c[4] - c[3], c[7] - c[4], c[5] - c[7], c[6] - c[5]);
    }
  }
  ctx->marg_general_count = 0;
  for (int w = 0; w < W; ++w)
    if (general[w] && mws[w].m > 0 && mws[w].n > 0) { mws[w].general = 1; ++ctx->marg_general_count; }
  if (ctx->marg_general_count > 0) {
    // rank-deficient Amm (or an over-sized problem): thresholded eigen pseudo-inverse of the full Amm in global memory
    scratch_total = 0;
    for (int w = 0; w < W; ++w) {
      MargWin &M = mws[w];
      if (!M.general) continue;
      M.scratch_off = (long long)scratch_total;
      const size_t T = (size_t)M.m + M.n;
      scratch_total += T * T + T + 3 * (size_t)M.m * M.m + (size_t)M.n * M.m + 3 * (size_t)M.n * M.n + 2 * M.n + 64;
    }
    if (d_scr.alloc(sizeof(double) * std::max<size_t>(1, scratch_total)) != hipSuccess ||
        hipMemcpy(d_mw.p, mws.data(), sizeof(MargWin) * W, hipMemcpyHostToDevice) != hipSuccess)
      return fail(VILO_ERR_HIP);
    have_flags = false;   // (this kernel writes status too: read again below)
    hipLaunchKernelGGL(k_marginalize, dim3(W), dim3(MT), 0, ctx->stream, bd, d_mw.as<MargWin>(), d_drop.as<int>(), max_l0, d_scr.as<double>(),
                       d_J0.as<double>(), d_r0.as<double>(), d_status.as<int>());
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "k_marginalize launch failed"; return fail(VILO_ERR_HIP); }
  }
  float marg_ms = 0.f;
  if (hipEventRecord(ctx->ev1, ctx->stream) != hipSuccess || hipEventSynchronize(ctx->ev1) != hipSuccess ||
      hipEventElapsedTime(&marg_ms, ctx->ev0, ctx->ev1) != hipSuccess)
    return fail(VILO_ERR_HIP);
  ctx->last_marg_ms = marg_ms;
  // Status first: a window whose result is unusable must not overwrite a pool slot either.
  std::vector<int> status(W, 0);
  int any_bad = 0;
  if (!have_flags) {
    if (bd.prep_bad) hipLaunchKernelGGL(k_marg_imu0_bad, dim3((W + 255) / 256), dim3(256), 0, ctx->stream, W, bd.prep_bad, bd.imu_skip, d_pbad.as<int>());
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(flags.data(), d_status.p, sizeof(int) * 3 * (size_t)W, hipMemcpyDeviceToHost) != hipSuccess) return fail(VILO_ERR_HIP);
  }
  for (int w = 0; w < W; ++w) status[w] = flags[w];
  // A preintegration covariance that is not positive definite has no sqrt_info (its bad pivots were replaced by 1 so that the arithmetic
  // stays finite): an IMU factor built on it would give a finite but meaningless prior. Only the factors that ENTER this marginalisation
  // count — MARGIN_OLD uses the factor of interval 0 alone (estimator.cpp:1271-1297), MARGIN_SECOND_NEW no IMU factor at all
  // (:1389-1410) —, like the reference, which would yield a valid prior whatever the other intervals look like. The flags are those of
  // the records in force: written by the preparation (batch creation / vilo_batch_prepare) or, with re-propagation, by the mode-0 pass
  // above (k_marg_imu0_bad folds them per window behind the marginalisation's kernels). Treated like a non-finite result: the window goes
  // on without a prior and the call reports VILO_ERR_NUMERIC.
  if (bd.prep_bad)
    for (int w = 0; w < W; ++w)
      if (!skip[w] && modes[w] == 0 && flags[2 * (size_t)W + w]) status[w] = 1;
  // windows with a prior pool leave J0 / r0 on the device (slot next_prior_slot); only their kept-block bookkeeping is host side
  std::vector<int> dst_slot(W, -1), src_slot(W, -1);
  bool any_host = false;
  for (int w = 0; w < W; ++w) {
    if (skip[w]) continue;
    if (refs && refs[w].prior_pool) {
      vilo_prior_pool *pl = refs[w].prior_pool;
      if (refs[w].next_prior_slot < 0 || refs[w].next_prior_slot >= pl->n) return VILO_ERR_BAD_ARG;
      if (keep_prior[w]) { if (refs[w].prior_slot != refs[w].next_prior_slot) src_slot[w] = refs[w].prior_slot; else continue; }
      else if (status[w] || mws[w].m == 0 || mws[w].n == 0) continue;   // (no prior comes out of this window: its slot is left alone, meta.valid cleared below)
      dst_slot[w] = refs[w].next_prior_slot;
    } else {
      any_host = true;
    }
  }
  if (refs && refs[0].prior_pool) {
    ArenaBuf d_dst{ctx, bt}, d_src{ctx, bt};
    if (d_dst.alloc(sizeof(int) * W) != hipSuccess || d_src.alloc(sizeof(int) * W) != hipSuccess ||
        hipMemcpy(d_dst.p, dst_slot.data(), sizeof(int) * W, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_src.p, src_slot.data(), sizeof(int) * W, hipMemcpyHostToDevice) != hipSuccess)
      return VILO_ERR_HIP;
    hipLaunchKernelGGL(k_prior_scatter, dim3(W), dim3(256), 0, ctx->stream, W, d_mw.as<MargWin>(), d_dst.as<int>(), d_src.as<int>(), d_J0.as<double>(), d_r0.as<double>(),
                       refs[0].prior_pool->dJ, refs[0].prior_pool->dr);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return VILO_ERR_HIP;
  }
  // priors that go back to host memory: through the context's reusable page-locked staging (a fresh pageable vector of W x 74 KB was
  // zero-filled, then filled again through the runtime's bounce buffer: most of a fleet's marginalisation call)
  const double *J0 = nullptr, *r0 = nullptr;
  if (any_host) {
    const size_t nJ = (size_t)W * VILO_MAX_PRIOR_DIM * VILO_MAX_PRIOR_DIM, nr = (size_t)W * VILO_MAX_PRIOR_DIM;
    double *hJ = (double *)vilo_host_stage(ctx, 7, sizeof(double) * (nJ + nr));
    if (!hJ) return fail(VILO_ERR_HIP);
    if (hipMemcpyAsync(hJ, d_J0.p, nJ * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(hJ + nJ, d_r0.p, nr * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
      return fail(VILO_ERR_HIP);
    J0 = hJ; r0 = hJ + nJ;
  }
  for (int w = 0; w < W; ++w) {
    const MargWin &M = mws[w];
    if (skip[w]) continue;
    const int mode = modes[w];
    const bool pooled = refs && refs[w].prior_pool;
    vilo_prior &p = pooled ? refs[w].prior_pool->meta[refs[w].next_prior_slot] : out[w];
    if (keep_prior[w]) {
      const vilo_prior &q = *vilo_win_prior(in[w], refs ? refs + w : nullptr);
      if (&p != &q) {
        int sum_g = 0;
        for (int k = 0; k < q.n_blocks; ++k) { p.block_id[k] = q.block_id[k]; p.block_size[k] = q.block_size[k]; p.block_idx[k] = q.block_idx[k]; sum_g += q.block_size[k]; }
        p.n = q.n; p.n_blocks = q.n_blocks; p.valid = 1;
        memmove(p.x0, q.x0, sizeof(double) * sum_g);
        if (!pooled) {
          memmove(p.J0, q.J0, sizeof(double) * (size_t)q.n * q.n);
          memmove(p.r0, q.r0, sizeof(double) * q.n);
        }
      }
      continue;
    }
    if (M.m == 0 || M.n == 0) { p.valid = 0; p.n = 0; continue; }
    // a non-finite result concerns this window alone: it goes on without a prior, the others keep theirs (the call still reports it)
    if (status[w]) { p.valid = 0; p.n = 0; any_bad = 1; continue; }
    const int WS = in[w].n_frames - 1;
    p.n = M.n; p.n_blocks = (int)kept_ids[w].size(); p.valid = 1;
    int xo = 0;
    const double *xs[6] = {state[w].pose, state[w].speed_bias, state[w].leg_bias, state[w].ex_pose, state[w].td, nullptr};
    for (int k = 0; k < p.n_blocks; ++k) {
      int id = kept_ids[w][k];
      const int kind = id / 16, index = id % 16;
      const double *src = kind == VILO_BLK_POSE ? xs[0] + 7 * index : kind == VILO_BLK_SB ? xs[1] + 9 * index : kind == VILO_BLK_LB ? xs[2] + 4 * index : kind == VILO_BLK_EX ? xs[3] + 7 * index : xs[4];
      // addr_shift: MARGIN_OLD frame k -> k-1 (estimator.cpp:1358-1368); MARGIN_SECOND_NEW frame WS -> WS-1 (:1413-1447)
      if (kind <= VILO_BLK_LB) {
        if (mode == 0) id = kind * 16 + (index - 1);
        else if (index == WS) id = kind * 16 + (index - 1);
      }
      p.block_id[k] = id; p.block_size[k] = kept_gs[w][k]; p.block_idx[k] = kept_cd[w][k];
      for (int c = 0; c < kept_gs[w][k]; ++c) p.x0[xo + c] = src[c];
      xo += kept_gs[w][k];
    }
    if (pooled) continue;
    memcpy(p.J0, J0 + (size_t)w * VILO_MAX_PRIOR_DIM * VILO_MAX_PRIOR_DIM, sizeof(double) * (size_t)M.n * M.n);   // (n x n packed on both sides)
    memcpy(p.r0, r0 + (size_t)w * VILO_MAX_PRIOR_DIM, sizeof(double) * M.n);
  }
  if (any_bad) { ctx->err = "non-finite marginalisation result (the windows concerned continue without a prior)"; return VILO_ERR_NUMERIC; }
  return VILO_OK;
}

extern "C" int vilo_marginalize(vilo_ctx *ctx, int W, const vilo_window_desc *in, const vilo_window_state *state, int mode,
                                vilo_prior *out) {
  if (!ctx || W <= 0 || !in || !state || !out || (mode != 0 && mode != 1)) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  vilo_batch *bt = nullptr;
  int rc = vilo_batch_create(ctx, W, in, state, &bt);
  if (rc != VILO_OK) return rc;
  std::vector<int> modes(W, mode);
  rc = marginalize_batch(ctx, bt, W, in, nullptr, state, modes.data(), out);
  vilo_batch_destroy(ctx, bt);
  return rc;
}

// The marginalisation half on a batch that is already resident (after vilo_batch_solve + vilo_batch_download): linearised at the batch's
// device state; `state` is the host copy of it (keep_block_data of the new prior). modes[w]: 0 MARGIN_OLD, 1 MARGIN_SECOND_NEW, < 0 skip.
extern "C" int vilo_batch_marginalize(vilo_ctx *ctx, vilo_batch *bt, int W, const vilo_window_desc *in, const vilo_window_state *state, const int *modes,
                                      vilo_prior *out) {
  if (!ctx || !bt || W <= 0 || W != vilo_batch_dev(bt)->W || !in || !state || !modes || !out) return VILO_ERR_BAD_ARG;
  for (int w = 0; w < W; ++w)
    if (modes[w] > 1) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  return marginalize_batch(ctx, bt, W, in, nullptr, state, modes, out);
}

// Estimator::optimization() (estimator.cpp:1054-1458) as one call on one device-resident batch: ceres::Solve, double2vector's
// gauge fix, then the marginalisation linearised at that result.
extern "C" int vilo_optimize_windows_resident(vilo_ctx *ctx, int W, const vilo_window_desc *in, const vilo_resident_refs *refs, vilo_window_state *inout,
                                              const vilo_solve_opts *opts, const int *marginalization_flag, vilo_prior *next_prior,
                                              vilo_solve_summary *summaries) {
  if (!ctx || W <= 0 || !in || !inout || !opts) return VILO_ERR_BAD_ARG;
  if (marginalization_flag && !next_prior) {
    for (int w = 0; w < W; ++w)
      if (!refs || !refs[w].prior_pool) return VILO_ERR_BAD_ARG;
  }
  VILO_HIP(hipSetDevice(ctx->device));
  const bool timing = getenv("VILO_HOST_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  vilo_batch *bt = nullptr;
  int rc = vilo_batch_create_refs(ctx, W, in, refs, inout, &bt);
  if (rc != VILO_OK) return rc;
  const double t1 = now();
  rc = vilo_batch_solve(ctx, bt, opts);
  const double t2 = now();
  if (rc == VILO_OK) rc = vilo_gauge_fix_batch(ctx, *vilo_batch_dev(bt));
  if (rc == VILO_OK) rc = vilo_batch_download(ctx, bt, inout, summaries);
  const double t3 = now();
  if (rc == VILO_OK && marginalization_flag) {
    // the reference only marginalises full windows (estimator.cpp:1243-1244)
    std::vector<int> modes(W);
    for (int w = 0; w < W; ++w) modes[w] = (in[w].n_frames == VILO_MAX_FRAMES) ? marginalization_flag[w] : -1;
    rc = marginalize_batch(ctx, bt, W, in, refs, inout, modes.data(), next_prior);
  }
  const double t4 = now();
  vilo_batch_destroy(ctx, bt);
  if (timing)
    fprintf(stderr, "[vilo_optimize_windows] W=%d create %.2f ms, solve %.2f ms, gauge fix + download %.2f ms, marginalise %.2f ms (kernels %.2f), destroy %.2f ms\n", W,
            t1 - t0, t2 - t1, t3 - t2, t4 - t3, ctx->last_marg_ms, now() - t4);
  return rc;
}

// Estimator::optimization() (estimator.cpp:1054-1458) as one call on one device-resident batch: ceres::Solve, double2vector's
// gauge fix, then the marginalisation linearised at that result.
extern "C" int vilo_optimize_windows(vilo_ctx *ctx, int W, const vilo_window_desc *in, vilo_window_state *inout, const vilo_solve_opts *opts,
                                     const int *marginalization_flag, vilo_prior *next_prior, vilo_solve_summary *summaries) {
  if (marginalization_flag && !next_prior) return VILO_ERR_BAD_ARG;
  if (!ctx || W <= 0 || !in || !inout || !opts) return VILO_ERR_BAD_ARG;
  // many host windows: sub-batches over the context's pipeline lanes (vilo_set_host_pipeline), like vilo_solve_windows — each window's
  // solve, gauge fix and marginalisation are its own, so the call's results are the one batch's
  int rc = VILO_OK;
  if (vilo_run_on_lanes(ctx, W, in, inout, [&](vilo_ctx *l, int w0, int n) {
        return vilo_optimize_windows_resident(l, n, in + w0, nullptr, inout + w0, opts, marginalization_flag ? marginalization_flag + w0 : nullptr,
                                              next_prior ? next_prior + w0 : nullptr, summaries ? summaries + w0 : nullptr);
      }, &rc))
    return rc;
  return vilo_optimize_windows_resident(ctx, W, in, nullptr, inout, opts, marginalization_flag, next_prior, summaries);
}

// ---- prior pool ----
extern "C" int vilo_prior_pool_create(vilo_ctx *ctx, int n, vilo_prior_pool **out) {
  if (!ctx || n <= 0 || !out) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  vilo_prior_pool *pl = new vilo_prior_pool();
  pl->n = n; pl->device = ctx->device; pl->dJ = pl->dr = nullptr;
  if (hipMalloc((void **)&pl->dJ, sizeof(double) * (size_t)n * 96 * 96) != hipSuccess || hipMalloc((void **)&pl->dr, sizeof(double) * (size_t)n * 96) != hipSuccess) {
    if (pl->dJ) (void)hipFree(pl->dJ);
    delete pl;
    ctx->err = "vilo_prior_pool_create: allocation failed";
    return VILO_ERR_HIP;
  }
  pl->meta.resize(n);
  pl->x0_store.assign((size_t)n * 7 * VILO_MAX_PRIOR_BLOCKS, 0.0);
  for (int i = 0; i < n; ++i) {
    memset(&pl->meta[i], 0, sizeof(vilo_prior));
    pl->meta[i].x0 = &pl->x0_store[(size_t)i * 7 * VILO_MAX_PRIOR_BLOCKS];
  }
  *out = pl;
  return VILO_OK;
}
extern "C" void vilo_prior_pool_destroy(vilo_ctx *ctx, vilo_prior_pool *pl) {
  if (!pl) return;
  (void)ctx;
  (void)hipSetDevice(pl->device);
  (void)hipFree(pl->dJ);
  (void)hipFree(pl->dr);
  delete pl;
}
extern "C" int vilo_prior_pool_dim(const vilo_prior_pool *pl, int slot) {
  if (!pl || slot < 0 || slot >= pl->n || !pl->meta[slot].valid) return 0;
  return pl->meta[slot].n;
}
extern "C" int vilo_prior_pool_upload(vilo_ctx *ctx, vilo_prior_pool *pl, int slot, const vilo_prior *p) {
  if (!ctx || !pl || slot < 0 || slot >= pl->n) return VILO_ERR_BAD_ARG;
  vilo_prior &m = pl->meta[slot];
  if (!p || !p->valid || p->n <= 0) { m.valid = 0; m.n = 0; m.n_blocks = 0; return VILO_OK; }
  if (p->n > VILO_MAX_PRIOR_DIM || p->n_blocks > VILO_MAX_PRIOR_BLOCKS || !p->x0 || !p->J0 || !p->r0) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  double *x0 = m.x0;
  m = *p;
  m.x0 = x0; m.J0 = nullptr; m.r0 = nullptr;
  int sum_g = 0;
  for (int k = 0; k < p->n_blocks; ++k) sum_g += p->block_size[k];
  memcpy(m.x0, p->x0, sizeof(double) * sum_g);
  VILO_HIP(hipMemcpy(pl->dJ + (size_t)slot * 96 * 96, p->J0, sizeof(double) * (size_t)p->n * p->n, hipMemcpyHostToDevice));
  VILO_HIP(hipMemcpy(pl->dr + (size_t)slot * 96, p->r0, sizeof(double) * p->n, hipMemcpyHostToDevice));
  return VILO_OK;
}
extern "C" int vilo_prior_pool_download(vilo_ctx *ctx, vilo_prior_pool *pl, int slot, vilo_prior *out) {
  if (!ctx || !pl || !out || slot < 0 || slot >= pl->n) return VILO_ERR_BAD_ARG;
  const vilo_prior &m = pl->meta[slot];
  double *x0 = out->x0, *J0 = out->J0, *r0 = out->r0;
  if (!m.valid || m.n <= 0) { out->valid = 0; out->n = 0; out->n_blocks = 0; return VILO_OK; }
  if (!x0 || !J0 || !r0) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  *out = m;
  out->x0 = x0; out->J0 = J0; out->r0 = r0;
  int sum_g = 0;
  for (int k = 0; k < m.n_blocks; ++k) sum_g += m.block_size[k];
  memcpy(x0, m.x0, sizeof(double) * sum_g);
  VILO_HIP(hipMemcpy(J0, pl->dJ + (size_t)slot * 96 * 96, sizeof(double) * (size_t)m.n * m.n, hipMemcpyDeviceToHost));
  VILO_HIP(hipMemcpy(r0, pl->dr + (size_t)slot * 96, sizeof(double) * m.n, hipMemcpyDeviceToHost));
  return VILO_OK;
}
