// ceres::CostFunction-shaped batched factor evaluation on gfx950 (API-compatibility path: Jacobians are
// materialised in the reference's row-major global-size layout). The fused solver (kernels_solve.hip)
// reuses the same device functions from factors.hpp but never materialises Jacobians.
#include "vilo_internal.hpp"

using namespace vilo;

// -------------------------------------------------------------------------------------------------
// sqrt_info = LLT(cov^-1).matrixL()^T  (imu_leg_factor.cpp:197-198, imu_factor.h:73), hoisted out of the
// iteration loop: one wave per interval, computed once per solve. cov = M M^T with M upper triangular
// (Cholesky of the index-reversed matrix), sqrt_info = M^-1. No explicit inverse of cov is formed.
// -------------------------------------------------------------------------------------------------
// Register form: lane i owns row i of the index-reversed covariance (N <= 31 doubles per lane); pivots and factor entries reach the
// other lanes through v_readlane (SGPR broadcast, no LDS round trip, no barrier): right-looking Cholesky, then column c of M^-1 by back
// substitution in lane c with the factor broadcast entry by entry. The arithmetic (operands, order of the subtractions, sqrt and division
// of the pivots) is the one of a left-looking loop over LDS, which this replaces (k_prepare_preint 1.64 -> 1.17 ms per 40 960 records;
// a rolled form with a shifting register window — 3 KB of code instead of 36 KB — was slower, 1.50 ms: the pace is set by the
// v_readlane -> SGPR -> v_fma chains, twice as many of them without the triangular bounds, not by the instruction fetch).
__device__ __forceinline__ double bcast_lane(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src);
  hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ void sqrt_info_wave(const double *cov, double *U_out, double *R /*LDS >= N*N*/, double * /*Ub*/, int *status) {
  const int lane = threadIdx.x;
  const int row = lane < N ? lane : N - 1;
  for (int e = lane; e < N * N; e += 64) R[e] = cov[e];   // coalesced; lane = row reads stride N (odd): no bank conflicts
  __syncthreads();
  double a[N];
#pragma unroll
  for (int j = 0; j < N; ++j) a[j] = R[(N - 1 - row) * N + (N - 1 - j)];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double piv = bcast_lane(a[j], j);
    if (!(piv > 0.0) || !isfinite(piv)) { bad = true; piv = 1.0; }
    const double d = sqrt(piv);
    const double lj = (row == j) ? d : a[j] / d;   // (rows above the diagonal carry values nobody reads)
    a[j] = lj;
#pragma unroll
    for (int q = j + 1; q < N; ++q) {
      a[q] -= lj * bcast_lane(lj, q);
      if (((q - j) & 7) == 0) __builtin_amdgcn_sched_barrier(0);   // eight broadcasts (SGPR pairs) in flight at a time: hoisted together they spill
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (bad && lane == 0) *status = 1;
  // M(i,k) = L(N-1-i, N-1-k) (upper). Column c = lane of U = M^-1 by back substitution; U(k,c) = 0 below the diagonal, so every lane can
  // run the sum to the end (the extra terms are exact zeros)
#pragma unroll
  for (int j = 0; j < N; ++j) asm volatile("" : "+v"(a[j]));   // (opaque: else the broadcasts above are recognised and kept alive in SGPRs)
  double u[N];
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double s = (i == lane) ? 1.0 : 0.0;
    // (the source lane is opaque per row: as constants, instruction selection emits the 496 broadcasts of all rows up front — they depend
    // on nothing that changes here — and the register allocator spills every one of them)
    int src = N - 1 - i;
    asm volatile("" : "+s"(src));
#pragma unroll
    for (int k = i + 1; k < N; ++k) {
      s -= bcast_lane(a[N - 1 - k], src) * u[k];
      if (((k - i) & 7) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    double v = s / bcast_lane(a[N - 1 - i], src);
    asm volatile("" : "+v"(v));   // (pins this row's arithmetic before the next row's broadcasts)
    u[i] = (i <= lane) ? v : 0.0;
    __builtin_amdgcn_sched_barrier(0);
  }
  if (lane < N) {
#pragma unroll
    for (int i = 0; i < N; ++i) U_out[i * N + lane] = u[i];
  }
}

// Two records per wave, the factor broadcast through LDS (the form k_prepare_preint runs). The register form above leaves half a wave
// idle (31 rows) and pays two v_readlane per multiply-add; a v_readlane cannot serve two records (one scalar for 64 lanes), an LDS read
// whose address is uniform per half-wave can. Lanes 0 .. 30 and 32 .. 62 own the rows of two records; L is kept as a packed lower
// triangle (entry (r, c) at r (r + 1) / 2 + c: 496 doubles per record — triangular numbers are distinct modulo 32, so a column's
// write has no bank conflict), column j written by its rows and read back by all lanes for the trailing update, and the back
// substitution of M^-1 reads the rows of the same triangle. Same operands, same order, same sqrt and divisions as sqrt_info_wave:
// bitwise the same sqrt_info. The covariance is staged through the triangle's own 496 doubles in two passes (16 + 15 source rows).
__device__ __forceinline__ constexpr int tri_at(int r, int c) { return r * (r + 1) / 2 + c; }
// (a wave's LDS operations execute in issue order: only the compiler must keep it. The pointer is an operand because a clobber alone does
// not cover a __shared__ array whose address never escapes)
__device__ __forceinline__ void lds_order(double *p) { asm volatile("" : : "v"(p) : "memory"); }
// offset (doubles) in vilo_preint of the e-th of PreintHead's 126 values: the first 33 are the record's own leading scalars in order,
// the others entries of the Jacobian (fill_preint_head above is the same table as a loop)
__device__ __forceinline__ int head_src(int e) {
  if (e < 33) return e;
  int row, col;
  if (e < 78) {
    const int m = (e - 33) / 9, idx = (e - 33) % 9;
    row = (m < 2 ? 0 : (m == 2 ? 3 : 6)) + idx / 3;
    col = ((m == 0 || m == 3) ? 21 : 24) + idx % 3;
  } else if (e < 114) {
    const int j = (e - 78) / 9, idx = (e - 78) % 9;
    row = 9 + 3 * j + idx / 3;
    col = 24 + idx % 3;
  } else {
    const int j = (e - 114) / 3;
    row = 9 + 3 * j + (e - 114) % 3;
    col = 27 + j;
  }
  return 33 + row * 31 + col;
}
static_assert(offsetof(vilo_preint, jacobian) == 33 * sizeof(double) && offsetof(vilo::PreintHead, dp_dba) == 33 * sizeof(double), "head_src");

__global__ void __launch_bounds__(64) k_prepare_preint(int n, const vilo_preint *pre, PreintPrepared *out, int *status, const unsigned char *skip, int per_record) {
  constexpr int N = 31, NT = N * (N + 1) / 2, NA = 16 * N /* = NT: the first staging pass takes 16 source rows */;
  static_assert(NA == NT, "staging passes");
  __shared__ double Ls[2][NT];
  const int lane = threadIdx.x, half = lane >> 5, l = lane & 31;
  const int f = 2 * blockIdx.x + half;
  const bool live = f < n && !(skip && skip[f]);
  const unsigned long long lv = __builtin_amdgcn_ballot_w64(live);
  if (!lv) return;
  const int fs = live ? f : (f ^ 1);   // a half without a record computes along on the other half's and stores nothing
  const double *src = (const double *)(pre + fs);
  double *L = Ls[half];
  if (live) {
    double *hd = (double *)&out[f].head;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = l + 32 * q;
      if (e < 126) hd[e] = src[head_src(e)];
    }
  }
  const double *cov = src + (offsetof(vilo_preint, covariance) / sizeof(double));
  const int row = l < N ? l : N - 1, sr = N - 1 - row;   // lane = row of the index-reversed covariance = source row sr, read backwards
  double a[N];
  {
    double st[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) st[q] = cov[min(l + 32 * q, NT - 1)];
#pragma unroll
    for (int q = 0; q < 16; ++q) if (l + 32 * q < NT) L[l + 32 * q] = st[q];
#pragma unroll
    for (int q = 0; q < 15; ++q) st[q] = cov[NT + min(l + 32 * q, N * N - NT - 1)];
    lds_order(L);
    if (sr < 16) {
#pragma unroll
      for (int j = 0; j < N; ++j) a[j] = L[sr * N + (N - 1 - j)];
    }
    lds_order(L);
#pragma unroll
    for (int q = 0; q < 15; ++q) if (l + 32 * q < N * N - NT) L[l + 32 * q] = st[q];
    lds_order(L);
    if (sr >= 16) {
#pragma unroll
      for (int j = 0; j < N; ++j) a[j] = L[(sr - 16) * N + (N - 1 - j)];
    }
    lds_order(L);
  }
  bool bad = false;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double p0 = bcast_lane(a[j], j), p1 = bcast_lane(a[j], 32 + j);
    double piv = half ? p1 : p0;
    if (!(piv > 0.0) || !isfinite(piv)) { bad = true; piv = 1.0; }
    const double d = sqrt(piv);
    const double lj = (row == j) ? d : a[j] / d;
    if (l < N && l >= j) L[tri_at(row, j)] = lj;
    lds_order(L);
#pragma unroll
    for (int q = j + 1; q < N; ++q) {
      a[q] -= lj * L[tri_at(q, j)];
      asm volatile("" : "+v"(a[q]));   // (here and now: left to itself the compiler sinks every column's update to where the entry is next read — a left-looking loop with all of L in registers)
    }
    lds_order(L);
  }
  if (bad && live && l == 0) status[per_record ? f : 0] = 1;
  // M(i,k) = L(N-1-i, N-1-k) (upper). Column c = lane of U = M^-1 by back substitution, as in sqrt_info_wave
  double u[N];
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double s = (i == l) ? 1.0 : 0.0;
#pragma unroll
    for (int k = i + 1; k < N; ++k) s -= L[tri_at(N - 1 - i, N - 1 - k)] * u[k];
    double v = s / L[tri_at(N - 1 - i, N - 1 - i)];
    asm volatile("" : "+v"(v) : "v"(L) : "memory");   // (the next row's reads stay behind this row's arithmetic: hoisted together they spill)
    u[i] = (i <= l) ? v : 0.0;
  }
  if (live && l < N) {
    double *U_out = out[f].sqrt_info;
#pragma unroll
    for (int i = 0; i < N; ++i) U_out[i * N + l] = u[i];
  }
}

// The reference's route taken literally (imu_leg_factor.cpp:197-198: LLT(covariance.inverse()).matrixL().transpose()): the inverse by
// Gauss-Jordan elimination with partial pivoting (what Eigen's inverse() does for a 31 x 31 matrix up to the elimination order), then the
// lower Cholesky factor of it, transposed; selectable with vilo_set_sqrt_info_mode. The covariance's condition number of 1e13 .. 1e14 is
// units (variances of 1e-11 beside ones of 0.1; ~ 15 after diagonal equilibration): this route and the default one (sqrt_info_wave: no
// inverse of the covariance is formed) both give the exact matrix to a few 1e-15 row by row (tests/test_oracle_factors.py, 100 digits).
template <int N>
__device__ void sqrt_info_literal_wave(const double *cov, double *U_out, double *A /*LDS N*(N+1)*/, double *B /*LDS N*(N+1)*/, int *status) {
  const int lane = threadIdx.x;
  const int LD = N + 1;
  for (int e = lane; e < N * N; e += 64) {
    const int i = e / N, j = e % N;
    A[i * LD + j] = cov[e];
    B[i * LD + j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < N; ++j) {
    // pivot: the largest |A[i][j]|, i >= j (ties: the lowest row)
    double best = (lane >= j && lane < N) ? fabs(A[lane * LD + j]) : -1.0;
    int bi = lane;
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_down(best, off, 64);
      const int oi = __shfl_down(bi, off, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    const int p = __shfl(bi, 0, 64);
    if (p != j && lane < N) {
      double t = A[j * LD + lane]; A[j * LD + lane] = A[p * LD + lane]; A[p * LD + lane] = t;
      t = B[j * LD + lane]; B[j * LD + lane] = B[p * LD + lane]; B[p * LD + lane] = t;
    }
    __syncthreads();
    const double piv = A[j * LD + j];
    if (lane == 0 && (!(fabs(piv) > 0.0) || !isfinite(piv))) *status = 1;
    __syncthreads();
    if (lane < N) { A[j * LD + lane] /= piv; B[j * LD + lane] /= piv; }
    __syncthreads();
    if (lane < N && lane != j) {   // lane = row
      const double f = A[lane * LD + j];
      for (int c = 0; c < N; ++c) { A[lane * LD + c] -= f * A[j * LD + c]; B[lane * LD + c] -= f * B[j * LD + c]; }
    }
    __syncthreads();
  }
  // lower Cholesky factor of the inverse (Eigen's LLT reads the lower triangle), in place in B
  for (int j = 0; j < N; ++j) {
    double sacc = 0.0;
    if (lane >= j && lane < N) {
      sacc = B[lane * LD + j];
      for (int k = 0; k < j; ++k) sacc -= B[lane * LD + k] * B[j * LD + k];
    }
    __syncthreads();
    if (lane == j) {
      if (!(sacc > 0.0) || !isfinite(sacc)) { *status = 1; sacc = 1.0; }
      B[j * LD + j] = sqrt(sacc);
    }
    __syncthreads();
    if (lane > j && lane < N) B[lane * LD + j] = sacc / B[j * LD + j];
    __syncthreads();
  }
  for (int e = lane; e < N * N; e += 64) {
    const int i = e / N, j = e % N;
    U_out[e] = (j >= i) ? B[j * LD + i] : 0.0;
  }
}

// skip (optional, [n]): records that carry no factor (interval beyond the window, sum_dt > 10 s) are left alone. per_record: the
// "covariance not positive definite" flag goes to status[f] instead of status[0], so that a batch can fail the one window it concerns.
// (the literal route is a kernel of its own: it needs two LDS matrices, the default one a single staging copy — 8 KB instead of 16 KB
// per wave lets the register file, not LDS, set the occupancy)
__global__ void __launch_bounds__(64) k_prepare_preint_literal(int n, const vilo_preint *pre, PreintPrepared *out, int *status, const unsigned char *skip, int per_record) {
  __shared__ double R[31 * 32], Ub[31 * 32];
  const int f = blockIdx.x;
  if (f >= n || (skip && skip[f])) return;
  const vilo_preint &p = pre[f];
  if (threadIdx.x == 0) fill_preint_head(p, out[f].head);
  sqrt_info_literal_wave<31>(p.covariance, out[f].sqrt_info, R, Ub, status + (per_record ? f : 0));
}

__global__ void __launch_bounds__(64) k_prepare_preint_imu(int n, const vilo_preint_imu *pre, PreintPrepared *out, int *status, const unsigned char *skip, int per_record, int literal) {
  __shared__ double R[15 * 16], Ub[15 * 16];
  const int f = blockIdx.x;
  if (f >= n || (skip && skip[f])) return;
  const vilo_preint_imu &p = pre[f];
  if (threadIdx.x == 0) fill_preint_head_imu(p, out[f].head);
  // 15x15 sqrt_info stored in the leading 225 doubles
  if (literal) sqrt_info_literal_wave<15>(p.covariance, out[f].sqrt_info, R, Ub, status + (per_record ? f : 0));
  else sqrt_info_wave<15>(p.covariance, out[f].sqrt_info, R, Ub, status + (per_record ? f : 0));
}

int vilo_launch_prepare_preint(vilo_ctx *ctx, int n, const vilo_preint *d_pre, PreintPrepared *d_out, int *d_status, const unsigned char *d_skip, int per_record) {
  if (n <= 0) return VILO_OK;
  if (ctx->sqrt_info_mode) hipLaunchKernelGGL(k_prepare_preint_literal, dim3(n), dim3(64), 0, ctx->stream, n, d_pre, d_out, d_status, d_skip, per_record);
  else hipLaunchKernelGGL(k_prepare_preint, dim3((n + 1) / 2), dim3(64), 0, ctx->stream, n, d_pre, d_out, d_status, d_skip, per_record);
  VILO_HIP(hipGetLastError());
  return VILO_OK;
}
int vilo_launch_prepare_preint_imu(vilo_ctx *ctx, int n, const vilo_preint_imu *d_pre, PreintPrepared *d_out, int *d_status, const unsigned char *d_skip, int per_record) {
  if (n <= 0) return VILO_OK;
  hipLaunchKernelGGL(k_prepare_preint_imu, dim3(n), dim3(64), 0, ctx->stream, n, d_pre, d_out, d_status, d_skip, per_record, ctx->sqrt_info_mode);
  VILO_HIP(hipGetLastError());
  return VILO_OK;
}

// -------------------------------------------------------------------------------------------------
// Projection factors: one thread per residual block.
// -------------------------------------------------------------------------------------------------
__device__ inline void store_2x7(double *dst, const double *J6) {
  if (!dst) return;
  for (int r = 0; r < 2; ++r) {
    for (int c = 0; c < 6; ++c) dst[r * 7 + c] = J6[r * 6 + c];
    dst[r * 7 + 6] = 0.0;
  }
}

template <int KIND>
__global__ void __launch_bounds__(128) k_eval_proj(int n, double sq, const double *obs, const double *pose_i, const double *pose_j, const double *ex0,
                            const double *ex1, const double *inv_dep, const double *td, double *r, double *J_pi, double *J_pj,
                            double *J_e0, double *J_e1, double *J_l, double *J_td) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  double Ji[12], Jj[12], Je0[12], Je1[12], Jl[2], Jt[2], res[2];
  const bool want = J_pi || J_pj || J_e0 || J_e1 || J_l || J_td;
  proj_factor<KIND>(obs + 12 * f, KIND == 2 ? nullptr : pose_i + 7 * f, KIND == 2 ? nullptr : pose_j + 7 * f, ex0 + 7 * f,
                    KIND == 0 ? nullptr : ex1 + 7 * f, inv_dep[f], td[f], sq, res, want, Ji, Jj, Je0, Je1, Jl, Jt);
  r[2 * f] = res[0];
  r[2 * f + 1] = res[1];
  if (!want) return;
  if (KIND != 2) {
    store_2x7(J_pi ? J_pi + 14 * f : nullptr, Ji);
    store_2x7(J_pj ? J_pj + 14 * f : nullptr, Jj);
  }
  store_2x7(J_e0 ? J_e0 + 14 * f : nullptr, Je0);
  if (KIND != 0) store_2x7(J_e1 ? J_e1 + 14 * f : nullptr, Je1);
  if (J_l) { J_l[2 * f] = Jl[0]; J_l[2 * f + 1] = Jl[1]; }
  if (J_td) { J_td[2 * f] = Jt[0]; J_td[2 * f + 1] = Jt[1]; }
}

// -------------------------------------------------------------------------------------------------
// IMU(-leg) factor: one wave per factor. Raw Jacobian staged in LDS, whitened by the hoisted sqrt_info.
// -------------------------------------------------------------------------------------------------
template <int NRES, int NLOC, bool LEG>
__global__ void __launch_bounds__(64) k_eval_imu_family(int n, double g_norm, const PreintPrepared *pre, const double *pose_i,
                                                        const double *sb_i, const double *lb_i, const double *pose_j,
                                                        const double *sb_j, const double *lb_j, double *r, double *J0, double *J1,
                                                        double *J2, double *J3, double *J4, double *J5) {
  __shared__ double Jraw[NRES * NLOC];
  __shared__ double rraw[NRES];
  __shared__ double U[NRES * NRES];
  const int f = blockIdx.x;
  if (f >= n) return;
  const int lane = threadIdx.x;
  for (int e = lane; e < NRES * NLOC; e += 64) Jraw[e] = 0.0;
  for (int e = lane; e < NRES * NRES; e += 64) U[e] = pre[f].sqrt_info[e];
  __syncthreads();
  const bool want = J0 || J1 || J2 || J3 || J4 || J5;
  if (lane == 0) {
    if (LEG)
      imu_leg_raw(pre[f].head, g_norm, pose_i + 7 * f, sb_i + 9 * f, lb_i + 4 * f, pose_j + 7 * f, sb_j + 9 * f, lb_j + 4 * f, rraw,
                  want, Jraw, NLOC);
    else
      imu_raw(pre[f].head, g_norm, pose_i + 7 * f, sb_i + 9 * f, pose_j + 7 * f, sb_j + 9 * f, rraw, want, Jraw, NLOC);
  }
  __syncthreads();
  if (lane < NRES) {
    double s = 0.0;
    for (int k = lane; k < NRES; ++k) s += U[lane * NRES + k] * rraw[k];
    r[NRES * f + lane] = s;
  }
  if (!want) return;
  // local column -> (block, column-in-block, global size)
  constexpr int NBLK = LEG ? 6 : 4;
  const int loff[6] = {0, 6, 15, 19, 25, 34}, lsz[6] = {6, 9, 4, 6, 9, 4}, gsz[6] = {7, 9, 4, 7, 9, 4};
  const int ioff[4] = {0, 6, 15, 21}, isz[4] = {6, 9, 6, 9}, igsz[4] = {7, 9, 7, 9};
  double *outs[6] = {J0, J1, J2, J3, J4, J5};
  for (int b = 0; b < NBLK; ++b) {
    double *dst = outs[b];
    if (!dst) continue;
    const int lo = LEG ? loff[b] : ioff[b], ls = LEG ? lsz[b] : isz[b], gs = LEG ? gsz[b] : igsz[b];
    dst += (size_t)f * NRES * gs;
    for (int e = lane; e < NRES * gs; e += 64) {
      const int i = e / gs, c = e % gs;
      double s = 0.0;
      if (c < ls)
        for (int k = i; k < NRES; ++k) s += U[i * NRES + k] * Jraw[k * NLOC + lo + c];
      dst[e] = s;
    }
  }
}

// MarginalizationFactor::Evaluate: one block per evaluation.
__global__ void __launch_bounds__(128) k_eval_prior(int n_eval, int n, int n_blocks, const int *block_size, const int *block_idx,
                                                    const double *x0, const double *J0, const double *r0, const double *params,
                                                    int param_stride, double *r, double *J, int sum_gsize) {
  __shared__ double dx[VILO_MAX_PRIOR_DIM];
  const int e = blockIdx.x;
  if (e >= n_eval) return;
  const int t = threadIdx.x;
  if (t < n_blocks) {
    int off = 0;
    for (int b = 0; b < t; ++b) off += block_size[b];
    prior_dx(params + (size_t)e * param_stride + off, x0 + off, block_size[t], dx + block_idx[t]);
  }
  __syncthreads();
  for (int i = t; i < n; i += blockDim.x) {
    double s = r0[i];
    for (int k = 0; k < n; ++k) s += J0[(size_t)i * n + k] * dx[k];
    r[(size_t)e * n + i] = s;
  }
  if (J) {
    double *Je = J + (size_t)e * n * sum_gsize;
    int goff = 0;
    for (int b = 0; b < n_blocks; ++b) {
      const int gs = block_size[b], ls = gs == 7 ? 6 : gs, idx = block_idx[b];
      for (int q = t; q < n * gs; q += blockDim.x) {
        const int i = q / gs, c = q % gs;
        Je[(size_t)i * sum_gsize + goff + c] = (c < ls) ? J0[(size_t)i * n + idx + c] : 0.0;
      }
      goff += gs;
    }
  }
}

__global__ void k_pose_plus(int n, const double *x, const double *d, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  pose_plus(x + 7 * i, d + 6 * i, out + 7 * i);
}

// -------------------------------------------------------------------------------------------------
// C-ABI entry points (host pointers in, host pointers out).
// -------------------------------------------------------------------------------------------------
namespace {
struct Stage {
  vilo_ctx *ctx;
  std::vector<DevBuf *> bufs;
  ~Stage() { for (auto *b : bufs) delete b; }
  int up(const double *h, size_t n, double **d) {
    *d = nullptr;
    if (!h) return VILO_OK;
    DevBuf *b = new DevBuf();
    bufs.push_back(b);
    VILO_HIP(b->alloc(n * sizeof(double)));
    VILO_HIP(hipMemcpyAsync(b->p, h, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    *d = b->as<double>();
    return VILO_OK;
  }
  int out(const double *h, size_t n, double **d) {
    *d = nullptr;
    if (!h) return VILO_OK;
    DevBuf *b = new DevBuf();
    bufs.push_back(b);
    VILO_HIP(b->alloc(n * sizeof(double)));
    *d = b->as<double>();
    return VILO_OK;
  }
  int down(double *h, const double *d, size_t n) {
    if (!h) return VILO_OK;
    VILO_HIP(hipMemcpyAsync(h, d, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return VILO_OK;
  }
};
#define TRY(x) do { int rc_ = (x); if (rc_ != VILO_OK) return rc_; } while (0)
}  // namespace

template <int KIND>
static int eval_proj_impl(vilo_ctx *ctx, int n, const double *obs, const double *pose_i, const double *pose_j, const double *ex0,
                          const double *ex1, const double *inv_dep, const double *td, double *r, double *J_pi, double *J_pj,
                          double *J_e0, double *J_e1, double *J_l, double *J_td) {
  if (!ctx || n < 0 || !obs || !ex0 || !inv_dep || !td || !r) return VILO_ERR_BAD_ARG;
  if (n == 0) return VILO_OK;
  VILO_HIP(hipSetDevice(ctx->device));
  Stage S{ctx};
  double *d_obs, *d_pi, *d_pj, *d_e0, *d_e1, *d_l, *d_td, *d_r, *dJ[6];
  TRY(S.up(obs, 12 * (size_t)n, &d_obs));
  TRY(S.up(pose_i, 7 * (size_t)n, &d_pi));
  TRY(S.up(pose_j, 7 * (size_t)n, &d_pj));
  TRY(S.up(ex0, 7 * (size_t)n, &d_e0));
  TRY(S.up(ex1, 7 * (size_t)n, &d_e1));
  TRY(S.up(inv_dep, n, &d_l));
  TRY(S.up(td, n, &d_td));
  TRY(S.out(r, 2 * (size_t)n, &d_r));
  double *hJ[6] = {J_pi, J_pj, J_e0, J_e1, J_l, J_td};
  const size_t jsz[6] = {14, 14, 14, 14, 2, 2};
  for (int k = 0; k < 6; ++k) TRY(S.out(hJ[k], jsz[k] * n, &dJ[k]));
  const double sq = ctx->cfg.focal_length / 1.5;
  hipLaunchKernelGGL(k_eval_proj<KIND>, dim3((n + 127) / 128), dim3(128), 0, ctx->stream, n, sq, d_obs, d_pi, d_pj, d_e0, d_e1, d_l,
                     d_td, d_r, dJ[0], dJ[1], dJ[2], dJ[3], dJ[4], dJ[5]);
  VILO_HIP(hipGetLastError());
  TRY(S.down(r, d_r, 2 * (size_t)n));
  for (int k = 0; k < 6; ++k) TRY(S.down(hJ[k], dJ[k], jsz[k] * n));
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  return VILO_OK;
}

extern "C" int vilo_eval_proj2f1c(vilo_ctx *ctx, int n, const double *obs, const double *pose_i, const double *pose_j,
                                  const double *ex0, const double *inv_dep, const double *td, double *r, double *J_pose_i,
                                  double *J_pose_j, double *J_ex0, double *J_feat, double *J_td) {
  if (!pose_i || !pose_j) return VILO_ERR_BAD_ARG;
  return eval_proj_impl<0>(ctx, n, obs, pose_i, pose_j, ex0, nullptr, inv_dep, td, r, J_pose_i, J_pose_j, J_ex0, nullptr, J_feat, J_td);
}
extern "C" int vilo_eval_proj2f2c(vilo_ctx *ctx, int n, const double *obs, const double *pose_i, const double *pose_j,
                                  const double *ex0, const double *ex1, const double *inv_dep, const double *td, double *r,
                                  double *J_pose_i, double *J_pose_j, double *J_ex0, double *J_ex1, double *J_feat, double *J_td) {
  if (!pose_i || !pose_j || !ex1) return VILO_ERR_BAD_ARG;
  return eval_proj_impl<1>(ctx, n, obs, pose_i, pose_j, ex0, ex1, inv_dep, td, r, J_pose_i, J_pose_j, J_ex0, J_ex1, J_feat, J_td);
}
extern "C" int vilo_eval_proj1f2c(vilo_ctx *ctx, int n, const double *obs, const double *ex0, const double *ex1,
                                  const double *inv_dep, const double *td, double *r, double *J_ex0, double *J_ex1, double *J_feat,
                                  double *J_td) {
  if (!ex1) return VILO_ERR_BAD_ARG;
  return eval_proj_impl<2>(ctx, n, obs, nullptr, nullptr, ex0, ex1, inv_dep, td, r, nullptr, nullptr, J_ex0, J_ex1, J_feat, J_td);
}

extern "C" int vilo_eval_imu_leg(vilo_ctx *ctx, int n, const vilo_preint *pre, const double *pose_i, const double *sb_i,
                                 const double *lb_i, const double *pose_j, const double *sb_j, const double *lb_j, double *r,
                                 double *J_pose_i, double *J_sb_i, double *J_lb_i, double *J_pose_j, double *J_sb_j, double *J_lb_j) {
  if (!ctx || n < 0 || !pre || !pose_i || !sb_i || !lb_i || !pose_j || !sb_j || !lb_j || !r) return VILO_ERR_BAD_ARG;
  if (n == 0) return VILO_OK;
  VILO_HIP(hipSetDevice(ctx->device));
  Stage S{ctx};
  DevBuf d_pre, d_prep, d_status;
  VILO_HIP(d_pre.alloc(sizeof(vilo_preint) * (size_t)n));
  VILO_HIP(d_prep.alloc(sizeof(PreintPrepared) * (size_t)n));
  VILO_HIP(d_status.alloc(sizeof(int)));
  VILO_HIP(hipMemsetAsync(d_status.p, 0, sizeof(int), ctx->stream));
  VILO_HIP(hipMemcpyAsync(d_pre.p, pre, sizeof(vilo_preint) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  TRY(vilo_launch_prepare_preint(ctx, n, d_pre.as<vilo_preint>(), d_prep.as<PreintPrepared>(), d_status.as<int>(), nullptr, 0));
  double *d_in[6], *d_r, *dJ[6];
  const double *h_in[6] = {pose_i, sb_i, lb_i, pose_j, sb_j, lb_j};
  const size_t gs[6] = {7, 9, 4, 7, 9, 4};
  for (int k = 0; k < 6; ++k) TRY(S.up(h_in[k], gs[k] * n, &d_in[k]));
  TRY(S.out(r, 31 * (size_t)n, &d_r));
  double *hJ[6] = {J_pose_i, J_sb_i, J_lb_i, J_pose_j, J_sb_j, J_lb_j};
  for (int k = 0; k < 6; ++k) TRY(S.out(hJ[k], 31 * gs[k] * n, &dJ[k]));
  hipLaunchKernelGGL((k_eval_imu_family<31, 38, true>), dim3(n), dim3(64), 0, ctx->stream, n, ctx->cfg.g_norm,
                     d_prep.as<PreintPrepared>(), d_in[0], d_in[1], d_in[2], d_in[3], d_in[4], d_in[5], d_r, dJ[0], dJ[1], dJ[2],
                     dJ[3], dJ[4], dJ[5]);
  VILO_HIP(hipGetLastError());
  TRY(S.down(r, d_r, 31 * (size_t)n));
  for (int k = 0; k < 6; ++k) TRY(S.down(hJ[k], dJ[k], 31 * gs[k] * n));
  int status = 0;
  VILO_HIP(hipMemcpyAsync(&status, d_status.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  if (status) { ctx->err = "covariance not positive definite"; return VILO_ERR_NUMERIC; }
  return VILO_OK;
}

extern "C" int vilo_eval_imu(vilo_ctx *ctx, int n, const vilo_preint_imu *pre, const double *pose_i, const double *sb_i,
                             const double *pose_j, const double *sb_j, double *r, double *J_pose_i, double *J_sb_i,
                             double *J_pose_j, double *J_sb_j) {
  if (!ctx || n < 0 || !pre || !pose_i || !sb_i || !pose_j || !sb_j || !r) return VILO_ERR_BAD_ARG;
  if (n == 0) return VILO_OK;
  VILO_HIP(hipSetDevice(ctx->device));
  Stage S{ctx};
  DevBuf d_pre, d_prep, d_status;
  VILO_HIP(d_pre.alloc(sizeof(vilo_preint_imu) * (size_t)n));
  VILO_HIP(d_prep.alloc(sizeof(PreintPrepared) * (size_t)n));
  VILO_HIP(d_status.alloc(sizeof(int)));
  VILO_HIP(hipMemsetAsync(d_status.p, 0, sizeof(int), ctx->stream));
  VILO_HIP(hipMemcpyAsync(d_pre.p, pre, sizeof(vilo_preint_imu) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  TRY(vilo_launch_prepare_preint_imu(ctx, n, d_pre.as<vilo_preint_imu>(), d_prep.as<PreintPrepared>(), d_status.as<int>(), nullptr, 0));
  double *d_in[4], *d_r, *dJ[4];
  const double *h_in[4] = {pose_i, sb_i, pose_j, sb_j};
  const size_t gs[4] = {7, 9, 7, 9};
  for (int k = 0; k < 4; ++k) TRY(S.up(h_in[k], gs[k] * n, &d_in[k]));
  TRY(S.out(r, 15 * (size_t)n, &d_r));
  double *hJ[4] = {J_pose_i, J_sb_i, J_pose_j, J_sb_j};
  for (int k = 0; k < 4; ++k) TRY(S.out(hJ[k], 15 * gs[k] * n, &dJ[k]));
  hipLaunchKernelGGL((k_eval_imu_family<15, 30, false>), dim3(n), dim3(64), 0, ctx->stream, n, ctx->cfg.g_norm,
                     d_prep.as<PreintPrepared>(), d_in[0], d_in[1], (const double *)nullptr, d_in[2], d_in[3],
                     (const double *)nullptr, d_r, dJ[0], dJ[1], dJ[2], dJ[3], (double *)nullptr, (double *)nullptr);
  VILO_HIP(hipGetLastError());
  TRY(S.down(r, d_r, 15 * (size_t)n));
  for (int k = 0; k < 4; ++k) TRY(S.down(hJ[k], dJ[k], 15 * gs[k] * n));
  int status = 0;
  VILO_HIP(hipMemcpyAsync(&status, d_status.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  if (status) { ctx->err = "covariance not positive definite"; return VILO_ERR_NUMERIC; }
  return VILO_OK;
}

extern "C" int vilo_eval_prior(vilo_ctx *ctx, int n_eval, const vilo_prior *prior, const double *params, double *r, double *J) {
  if (!ctx || !prior || !params || !r || n_eval < 0) return VILO_ERR_BAD_ARG;
  if (!prior->valid || prior->n <= 0 || prior->n > VILO_MAX_PRIOR_DIM || prior->n_blocks > VILO_MAX_PRIOR_BLOCKS) return VILO_ERR_BAD_ARG;
  if (n_eval == 0) return VILO_OK;
  VILO_HIP(hipSetDevice(ctx->device));
  const int n = prior->n, nb = prior->n_blocks;
  int sum_g = 0;
  for (int b = 0; b < nb; ++b) sum_g += prior->block_size[b];
  Stage S{ctx};
  double *d_x0, *d_J0, *d_r0, *d_par, *d_r, *d_J;
  TRY(S.up(prior->x0, sum_g, &d_x0));
  TRY(S.up(prior->J0, (size_t)n * n, &d_J0));
  TRY(S.up(prior->r0, n, &d_r0));
  TRY(S.up(params, (size_t)sum_g * n_eval, &d_par));
  TRY(S.out(r, (size_t)n * n_eval, &d_r));
  TRY(S.out(J, (size_t)n * sum_g * n_eval, &d_J));
  DevBuf d_bs, d_bi;
  VILO_HIP(d_bs.alloc(sizeof(int) * nb));
  VILO_HIP(d_bi.alloc(sizeof(int) * nb));
  VILO_HIP(hipMemcpyAsync(d_bs.p, prior->block_size, sizeof(int) * nb, hipMemcpyHostToDevice, ctx->stream));
  VILO_HIP(hipMemcpyAsync(d_bi.p, prior->block_idx, sizeof(int) * nb, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_eval_prior, dim3(n_eval), dim3(128), 0, ctx->stream, n_eval, n, nb, d_bs.as<int>(), d_bi.as<int>(), d_x0, d_J0,
                     d_r0, d_par, sum_g, d_r, d_J, sum_g);
  VILO_HIP(hipGetLastError());
  TRY(S.down(r, d_r, (size_t)n * n_eval));
  TRY(S.down(J, d_J, (size_t)n * sum_g * n_eval));
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  return VILO_OK;
}

extern "C" int vilo_pose_plus(vilo_ctx *ctx, int n, const double *x, const double *delta, double *out) {
  if (!ctx || n < 0 || !x || !delta || !out) return VILO_ERR_BAD_ARG;
  if (n == 0) return VILO_OK;
  VILO_HIP(hipSetDevice(ctx->device));
  Stage S{ctx};
  double *d_x, *d_d, *d_o;
  TRY(S.up(x, 7 * (size_t)n, &d_x));
  TRY(S.up(delta, 6 * (size_t)n, &d_d));
  TRY(S.out(out, 7 * (size_t)n, &d_o));
  hipLaunchKernelGGL(k_pose_plus, dim3((n + 127) / 128), dim3(128), 0, ctx->stream, n, d_x, d_d, d_o);
  VILO_HIP(hipGetLastError());
  TRY(S.down(out, d_o, 7 * (size_t)n));
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  return VILO_OK;
}

extern "C" void vilo_huber(double delta, double s, double rho[3]) { vilo::huber_rho(delta, s, rho); }

// Calibration aid for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md, HBM section: the counters
// must be calibrated on a known byte count in the kernel's own access pattern): streams n doubles, 8 B per lane.
__global__ void k_calib_copy(const double *src, double *dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i] + 1.0;
}
extern "C" int vilo_debug_calib_copy(vilo_ctx *ctx, size_t n_doubles, int reps) {
  if (!ctx || n_doubles == 0) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  DevBuf a, b;
  VILO_HIP(a.alloc(n_doubles * 8));
  VILO_HIP(b.alloc(n_doubles * 8));
  VILO_HIP(hipMemsetAsync(a.p, 0, n_doubles * 8, ctx->stream));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_calib_copy, dim3(2048), dim3(256), 0, ctx->stream, a.as<double>(), b.as<double>(), n_doubles);
  VILO_HIP(hipGetLastError());
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  return VILO_OK;
}
