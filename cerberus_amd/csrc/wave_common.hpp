// Device helpers shared by the single-wave solver (kernels_wave.hip) and the eight-wave solver (kernels_mw8.hip): tile numbering of the
// 80 x 80 pose system (15 lower 16 x 16 tiles in FP64-MFMA accumulator order) and the 16 x 16 Cholesky + inverse of one diagonal tile.
#pragma once
#include "solve_common.hpp"

using namespace vilo;

#ifndef WS_C
#define WS_C 0   // panel slots of the blocked Cholesky start here in the workgroup's LDS (pswz)
#endif

__device__ __forceinline__ int tile_index(int I, int J) { return (I * (I + 1)) / 2 + J; }   // I >= J
__device__ constexpr int c_tI[15] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4};
__device__ constexpr int c_tJ[15] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4};

__device__ __forceinline__ int pswz_at(int base, int slot, int r, int c) { return base + 256 * slot + 16 * r + ((c + r) & 15); }
__device__ __forceinline__ int pswz(int slot, int r, int c) { return pswz_at(WS_C, slot, r, c); }

// 16 x 16 Cholesky + inverse of the factor by one wave (diagonal tile of the blocked 80 x 80 factorisation).
// A: LDS 16 x 17 row-major in. Lane i (< 16, replicated in the four 16-lane groups) owns row i; pivots broadcast with v_readlane.
// Writes L^-1 (lower, zeros above) to Linv (16 x 17): the panel products and the backward solve need L_jj^-1, nothing reads L_jj again.
// Returns 0 / 1 (not positive definite).
__device__ __forceinline__ int chol16_tile(const double *A, double *Linv) {
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
  // column c = lane of L^-1 by forward substitution; L is broadcast from the owning lanes' registers. (The copies are opaque to the
  // compiler: it would otherwise recognise these broadcasts as the ones of the factorisation loop and keep all 120 alive in SGPRs.)
#pragma unroll
  for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(a[j]));
  double cl[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    double v = (i == row) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < i; ++q) v -= readlane_d(a[q], i) * cl[q];   // L[i][q] lives in lane i, register q
    cl[i] = v * readlane_d(myrinv, i);
    // keep the v_readlane results (SGPR pairs) of one row at a time: the broadcasts depend on nothing that changes in this loop, so
    // instruction selection emits them all up front and they spill by the hundred unless every row's arithmetic is pinned in place
    asm volatile("" : "+v"(cl[i]));
    __builtin_amdgcn_sched_barrier(0);
  }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) Linv[i * 17 + lane] = (i >= lane) ? cl[i] : 0.0;
  }
  lds_fence();
  return fail;
}

