// The block-tridiagonal Cholesky chain of the speed / leg-bias part by one wave, with its coupling rows handed on through global memory:
//   S_k = A_kk + mu D_k - T_A(k+1)^T T_A(k+1),  L_k = chol(S_k),  M_k = L_k^-1,  T_A(k) = M_k A_{k,k-1},
//   V = [B_k | g_k] - T_A(k+1)^T T(k+1),  T(k) = M_k V   (13 x 80: columns 0..78 coupling rows, column 79 = rhs),  rhs_P -= T_B(k)^T t_g(k)
// for frames F-1 .. 0. The scalar part runs lane = row in the four 16-lane groups; T and V are FP64-MFMA tiles whose accumulator layout
// (register r of lane (lr, lk) = row lk + 4 r, column lr) is the operand layout of the next product — and of C -= T_B(k)^T T_B(k), which
// the caller does from Tout: [frame k][tile X][register kk][lane], tiles left of x_lo(k) neither written nor to be read.
// Used by k_chain (kernels_split.hip: two waves per SIMD) and by k_solve_wave's complete path (kernels_wave.hip).
#pragma once
#include "wave_common.hpp"

// scratch (doubles, 704)
#define CH_LM 0         // 13 x 13: M_k = L_k^-1
#define CH_TA0 176
#define CH_TA1 352
#define CH_SN 528       // 13 x 13: S_{k-1} = A_{k-1,k-1} - T_A(k)^T T_A(k)
#define CH_TOTAL 704

__device__ __forceinline__ int chain_x_lo(int k, int kb) { return (k <= kb) ? 0 : max(0, (6 * (k - 1)) >> 4); }   // T(k) is zero left of pose k - 1 (dense from the prior's frame down)

// DB / GB: dhat^2 and gradient of the 143 speed / leg-bias dimensions (LDS). Returns 1 when a pivot was not positive.
__device__ __forceinline__ int chain_to_global(const double *bimg, const double *gin, double *scr, const double *DB, const double *GB,
                                               double *Mg, double *TAg, double *Tout, double *vout, int F, int kb, double mu, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  int fX[5], oX[5];
#pragma unroll
  for (int X = 0; X < 5; ++X) { const int col = 16 * X + lr; fX[X] = col < 66 ? col / 6 : 99; oX[X] = col < 66 ? col - 6 * fX[X] : 0; }
  int fail = 0;
  {
    const int grp = lk, c = lr;
    const int row = c < 13 ? c : 0;
    double *LM = scr + CH_LM, *SN = scr + CH_SN;
    double *TAcur = scr + CH_TA0, *TAprev = scr + CH_TA1;
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
      const int x_lo = chain_x_lo(k, kb);
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
      // T_B(k) out in operand order (tiles left of x_lo are zero and skipped on both sides): the middle stage's C -= T_B^T T_B;
      // rhs_P -= T_B^T t_g here
      {
        double *Tb = Tout + 1280 * k;
#pragma unroll
        for (int X = 0; X < 5; ++X)
          if (X >= x_lo) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
              if (kk < 3 || lk == 0) Tb[(X * 4 + kk) * 64 + lane] = (X == 4) ? T4[kk] : T[X][kk];   // (rows 13 .. 15 of the 16: padding, never read)
          }
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
      if (lk == 0) vout[16 * X + lr] = gin[16 * X + lr] - yr[X];
    }
  }
  return fail;
}
