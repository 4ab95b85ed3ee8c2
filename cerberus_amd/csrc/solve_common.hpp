// Small device helpers shared by the solver kernels (kernels_solve.hip: linearisation, cost, accept; kernels_wave.hip: pose-system
// assembly and the single-wave solver; kernels_marg.hip reads the Gram slots through gram26_index).
#pragma once
#include "solver_types.hpp"

typedef double mfma_d4 __attribute__((ext_vector_type(4)));

// Shader-clock phase stamps (SolverState::phase_clk, read by tools/phase_clocks_wave.py) exist in a profiling build only
// (hipcc -DVILO_PHASE_CLOCKS: `VILO_BUILD_PROF=1 python __graft_entry__.py` -> lib/libvilo_gpu_prof.so, loaded through VILO_GPU_LIB);
// the production kernels carry none.
#ifdef VILO_PHASE_CLOCKS
#define PCLK(...) do { __VA_ARGS__; } while (0)
#define pclk64() clock64()
#else
#define PCLK(...) do { } while (0)
#define pclk64() 0LL
#endif

// LDS-only workgroup barrier: waits for this wave's LDS traffic (lgkmcnt) but leaves global loads in flight
// (HIP's __syncthreads() also drains vmcnt, which serialises every prefetch behind the barrier).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// single-wave workgroups: LDS writes of this wave are visible to its later reads once lgkmcnt has drained
// image[e] += v in LDS as ONE instruction (ds_add_f64, no return value): an owner thread's read-modify-writes of the assembly scatter are then
// a stream of independent LDS operations instead of a chain of read -> add -> write round trips. A target still has exactly one owner thread and
// a wave's LDS operations execute in program order, so the sums keep their fixed order (batch of N == batch of 1, bitwise).
__device__ __forceinline__ void lds_add(double *p, double v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// broadcast lane `src` (wave-uniform index) of a double through SGPRs (v_readlane): a few cycles, no LDS round trip
__device__ __forceinline__ double readlane_d(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src);
  hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}

// 1 / sqrt(x) as the device library forms it for a double — v_rsq_f64 and one Newton-like correction, five dependent operations —, spelled
// out so that chol_rows can place the operations one by one between other work.
__device__ __forceinline__ bool pos_finite(double x) { return __builtin_amdgcn_class(x, 0x180); }   // positive normal or subnormal
__device__ __forceinline__ double rsqrt_steps(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double t = y0 * -x, e = __builtin_fma(t, y0, 1.0), u = y0 * e, w = __builtin_fma(e, 0.375, 0.5), y = __builtin_fma(u, w, y0);
  return pos_finite(y0) ? y : y0;
}

// Right-looking Cholesky of an N x N matrix (N <= 16) in registers, lane = row (a[q]: entry (row, q) in, L(row, q) out; the four 16-lane
// groups of a wave may replicate the work), pivots and column entries broadcast with v_readlane; myrinv: 1 / L(row, row). A column step
// is the pivot's dependent chain — v_readlane -> class check -> v_rsq_f64 + five Newton operations -> scale: ~240 cycles — followed by the
// trailing update (a v_readlane pair + multiply-add per remaining column, ~24 cycles each). The pivot of the NEXT column is therefore
// formed one step ahead as a wave-uniform scalar: piv' = a(q+1, q+1) - (a(q+1, q) rinv)^2, the very multiply-add the vector update
// performs in lane q + 1, so that its reciprocal square root does not wait for the whole update. (Placing the chain's ten operations one
// by one between the updates, pinned with sched_barrier, measured slower: 6.7 k against 6.1 k cycles for 16 x 16.)
// The same operations on the same values as the plain loop (readlane the pivot after the update): bitwise the same factor.
// Returns 1 if a pivot was not positive and finite (it is replaced by 1, as Eigen's LLT would go on with garbage; the caller retries).
template <int N>
__device__ __forceinline__ int chol_rows(double (&a)[N], int row, double &myrinv) {
  int fail = 0;
  double piv = readlane_d(a[0], 0);
  if (!pos_finite(piv)) { fail = 1; piv = 1.0; }
  double rinv = rsqrt_steps(piv);
#pragma unroll
  for (int q = 0; q < N; ++q) {
    double pivn = 1.0, rinvn = 1.0;
    if (q + 1 < N) {
      const double a10 = readlane_d(a[q], q + 1), a11 = readlane_d(a[q + 1 < N ? q + 1 : q], q + 1);
      const double l10 = a10 * rinv;
      pivn = __builtin_fma(-l10, l10, a11);
      if (!pos_finite(pivn)) { fail = 1; pivn = 1.0; }
      rinvn = rsqrt_steps(pivn);
    }
    const double lq = (row == q) ? piv * rinv : (row > q ? a[q] * rinv : 0.0);
    a[q] = lq;
    if (row == q) myrinv = rinv;
    // trailing update, four columns at a time: the four broadcasts first, then the four multiply-adds (one v_readlane pair directly in front
    // of its multiply-add waits for the scalar registers every time: ~26 cycles per column instead of ~14)
#pragma unroll
    for (int q0 = q + 1; q0 < N; q0 += 4) {
      double sb[4];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) if (q0 + u < N) sb[u] = readlane_d(lq, q0 + u);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) if (q0 + u < N) a[q0 + u] = __builtin_fma(-lq, sb[u], a[q0 + u]);
    }
    __builtin_amdgcn_sched_barrier(0);
    piv = pivn; rinv = rinvn;
  }
  return fail;
}

__device__ __forceinline__ int tri26(int a, int b) { return a * 26 - (a * (a - 1)) / 2 + (b - a); }  // a <= b
__device__ __forceinline__ int tri23(int a, int b) { return a * 23 - (a * (a - 1)) / 2 + (b - a); }  // a <= b
// Entry (c1, c2) of the 26-column view [pose_s 6 | pose_j 6 | ex0 6 | ex1 6 | td | r] of a Gram slot: index into the packed 23-column
// slot and the sign it carries (the pose_j translation columns are minus the pose_s ones).
__device__ __forceinline__ int gram_col(int c, int &sg) {
  sg = 1;
  if (c < 6) return c;
  if (c < 9) { sg = -1; return GC_T + (c - 6); }
  if (c < 12) return GC_RJ + (c - 9);
  if (c < 18) return GC_E0 + (c - 12);
  if (c < 24) return GC_E1 + (c - 18);
  return c == 24 ? GC_TD : GC_R;
}
__device__ __forceinline__ int gram26_index(int c1, int c2, double &sign) {
  int s1, s2;
  const int m1 = gram_col(c1, s1), m2 = gram_col(c2, s2);
  sign = (double)(s1 * s2);
  return tri23(min(m1, m2), max(m1, m2));
}
__device__ __forceinline__ int tri39(int a, int b) { return a * 39 - (a * (a - 1)) / 2 + (b - a); }   // a <= b

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return __shfl(v, 0, 64);
}
__device__ __forceinline__ double wave_max(double v) {
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return __shfl(v, 0, 64);
}

// The same sums / maxima through the DPP data path (v_mov_b32_dpp: a VALU move with a lane pattern, a few cycles) instead of twelve
// ds_bpermute round trips through the LDS crossbar (~ 900 cycles per double, one after the other): row_shr 1 / 2 / 4 / 8 leave every row's
// total in its lane 15, row_bcast:15 / :31 carry the rows' totals on to lane 63, which is broadcast. A different (fixed) order of the
// additions than wave_sum's tree: used where no other kernel has to reproduce the sum bit for bit.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_move_d<0x111, 0xf>(v);
  v += dpp_move_d<0x112, 0xf>(v);
  v += dpp_move_d<0x114, 0xf>(v);
  v += dpp_move_d<0x118, 0xf>(v);
  v += dpp_move_d<0x142, 0xa>(v);
  v += dpp_move_d<0x143, 0xc>(v);
  return readlane_d(v, 63);
}
__device__ __forceinline__ double wave_max_dpp(double v) {   // of non-negative values (a lane without a source contributes 0)
  v = fmax(v, dpp_move_d<0x111, 0xf>(v));
  v = fmax(v, dpp_move_d<0x112, 0xf>(v));
  v = fmax(v, dpp_move_d<0x114, 0xf>(v));
  v = fmax(v, dpp_move_d<0x118, 0xf>(v));
  v = fmax(v, dpp_move_d<0x142, 0xa>(v));
  v = fmax(v, dpp_move_d<0x143, 0xc>(v));
  return readlane_d(v, 63);
}

struct SolveParams {
  double min_lm_diagonal, max_lm_diagonal;
  double min_radius, gradient_tolerance;
  int jacobi_scaling, fixed_iterations;
};

// is camera dimension cd part of the problem (not a constant block, not beyond the window's frames, not padding)
__device__ __forceinline__ bool cd_active(int cd, int F, int cmask) {
  if (cd < 66) return cd < 6 * F;
  if (cd < CD_TD) return !(cmask & CONST_EX);
  if (cd == CD_TD) return !(cmask & CONST_TD);
  if (cd < CD_B0 || cd >= CD_B0 + 143) return false;
  const int e = cd - CD_B0, k = e / 13, c = e - 13 * k;
  return k < F && !(c >= 9 && (cmask & CONST_LB));
}

// DoglegStrategy::ComputeTraditionalDoglegStep in scalar form + TrustRegionMinimizer's model_cost_change.
// (S: SolverState, or a register copy of the fields it reads and writes)
template <class S>
__device__ __forceinline__ void dogleg_scalars(S &s) {
  const double gradient_norm = sqrt(s.gnorm2), gn_norm = sqrt(s.gnnorm2), radius = s.radius;
  double a, bb, step_norm;
  if (gn_norm <= radius) {
    a = 0.0; bb = 1.0; step_norm = gn_norm;
  } else if (gradient_norm * s.alpha >= radius) {
    a = radius / gradient_norm; bb = 0.0; step_norm = radius;
  } else {
    const double b_dot_a = -s.alpha * s.gdotgn;
    const double a_squared_norm = (s.alpha * gradient_norm) * (s.alpha * gradient_norm);
    const double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + gn_norm * gn_norm;
    const double c = b_dot_a - a_squared_norm;
    const double d = sqrt(c * c + b_minus_a_squared_norm * (radius * radius - a_squared_norm));
    const double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + c);
    a = s.alpha * (1.0 - beta); bb = beta;
    step_norm = sqrt(fmax(0.0, a * a * s.gnorm2 - 2.0 * a * bb * s.gdotgn + bb * bb * s.gnnorm2));
  }
  s.coef_a = a; s.coef_b = bb; s.dogleg_step_norm = step_norm;
  // -(J d)^T (r + J d / 2) with d = -a D^-2 g - b y, (H + mu D^2) y = g
  const double gy = -s.gdotgn;
  s.model_cost_change = a * s.gnorm2 + bb * gy -
                        0.5 * (a * a * s.q + 2.0 * a * bb * (s.gnorm2 - s.mu * gy) + bb * bb * (gy - s.mu * s.gnnorm2));
  s.step_valid = (s.model_cost_change > 0.0) ? 1 : 0;
}
