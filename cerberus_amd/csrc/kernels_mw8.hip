// Eight-wave solver for gfx950, the form of batches that leave most of the chip idle (one workgroup = one window per CU; used up to 512
// windows — VILO_MW8_MAX_WINDOWS — in a second round of workgroups from 257 on): what ceres::Solve does per linearisation for Estimator::optimization(), estimator.cpp:1221-1236 — DENSE_SCHUR + traditional
// DOGLEG, Ceres 1.14 semantics. At this size a solve is ONE latency chain per window, so the window's work is cut into roles that run side
// by side, two waves per SIMD at 256 registers each (the four-wave form this replaces ran its serial parts on one wave each):
//
//   waves C1 / C2   the block-tridiagonal Cholesky of the speed / leg-bias part as a TWISTED factorisation: C1 eliminates frames F-1 .. m+1
//                   downwards, C2 frames 0 .. m-1 upwards, at the same time; the middle frame m = (F-1)/2 takes the Schur updates of both
//                   neighbours and is factorised last (by C1). Only the 13 x 13 part — S_k, L_k, M_k = L_k^-1 and the off-diagonal factor:
//                   the v_readlane chains that nothing shortens. The back-substitution sweeps run the same way: both halves forward, the
//                   middle, both halves backward. M_k and the off-diagonal factors stay in LDS (no round trip through L2).
//                     downwards:  T_A(k)  = L_k^-1 A_{k,k-1}      S_{k-1} = A_{k-1,k-1} - T_A(k)^T T_A(k)
//                     upwards:    T'_A(k) = L_k^-1 A_{k+1,k}^T    S_{k+1} = A_{k+1,k+1} - T'_A(k)^T T'_A(k)
//                     middle:     S_m = A_mm - T_A(m+1)^T T_A(m+1) - T'_A(m-1)^T T'_A(m-1)
//   waves TD / TU   the coupling rows of the chain on the matrix cores, one step behind their chain wave (M_k / T_A(k) come through LDS):
//                     downwards:  T(k) = M_k (V_k - T_A(k+1)^T T(k+1))      upwards:  T(k) = M_k (V_k - T'_A(k-1)^T T(k-1))
//                     middle (TD): V_m loses both neighbours' terms
//                   T(k) goes to the matrix waves through a double-buffered LDS image in operand order.
//   waves B1 .. B4  the 80 x 80 pose system split by tiles (4 + 4 + 4 + 3 of the 15 lower 16 x 16 tiles): each runs the landmark Schur
//                   complement and the rank updates C -= T(k)^T T(k) on its own tiles, two steps behind the chain (operands from LDS slices
//                   that the chain and T waves fetch one iteration ahead); all four share the landmark back-substitution.
//   all eight       the Cholesky of the 80 x 80 pose system (mw8_chol80): one wave carries the critical path — diagonal tile, its next
//                   block row, the next diagonal tile's update — the others the rest of the panel and the trailing tiles beside it.
// Every role is a __noinline__ function: the register allocator then sees one role at a time (as one kernel the roles spilled 46 .. 533
// registers inside their loops). Three things a role that is not a kernel needs, each measured: its pointer arguments typed as global
// memory (uni_g), the dynamic LDS base pinned in a register (e_lds_base), tables read at uniform addresses through a vector base
// (lds_in_vgpr) — see there.
// The chain step (13 x 13 factorisation + substitutions) sets the pace of the main loop; in the four-wave form a chain wave also formed
// T(k) (twice the cycles per frame) and two matrix waves shared the 15 tiles.
// A different elimination order of the speed / leg-bias part and different partial sums than the single-wave forms: results agree with
// them to rounding (tests run every form against the oracle at the same tolerances); batch-of-N == batch-of-1 bitwise within the form.
#include <type_traits>
#include "wave_common.hpp"

// LDS map (doubles)
#define E_G 0          // [80]  gradient of the pose part
#define E_DH2 80       // [80]  dogleg diagonal
#define E_Y 160        // [80]  Gauss-Newton step of the pose part
#define E_V 240        // [80]  reduced right-hand side
#define E_VP 320       // [80]  v_P = D^-2 g
#define E_DB 400       // [144] dogleg diagonal of the speed / leg-bias part
#define E_GB 544       // [144] its gradient
#define E_RED 688      // [96]  cross-wave sums and flags
#define E_CH1 784      // [704] C1's chain scratch
#define E_CH2 1488     // [704] C2's
#define E_XSN 2192     // [176] T'_A(m-1)^T T'_A(m-1): C2's update of the middle frame's diagonal block
#define E_XU 2368      // [16]  u of frame m - 1 (forward sweep hand-over)
#define E_XY 2384      // [16]  y of frame m (backward sweep hand-over)
#define E_M 2400       // [11][176] M_k = L_k^-1
#define E_TA 4336      // [11][176] T_A(k) (frames above the middle), T'_A(k) (frames below)
#define E_T1 6272      // [2][1280] TD's T(k) hand-over
#define E_T2 8832      // [2][1280] TU's
#define E_TILES E_T1   // after the main loop, over the hand-over buffers: [15][256] the tiles on their way to the Cholesky (accumulator order)
#define E_DEL 11392    // [224] the step
#define E_U 11616      // [144]
#define E_YB 11760     // [144]
#define E_SKIP 11904   // [40]
#define E_CHOL 11944   // Cholesky scratch (832, padded to 1024) + panel slots (1024)
#define E_WS 13992     // [2][ES_N] slices of the landmarks' coupling rows, 1 / (E + mu dhat^2) and gradients (mw8_role_T loads them)
#define ES_NL 32       // landmarks per slice (four half trips)
#define ES_LD 33       // row stride of a slice (16 rows x 4 landmarks of an operand read: distinct banks up to two-way)
#define ES_EI (80 * ES_LD)         // [32] 1 / (E_l + mu dhat_l^2), zero beyond the window's landmarks
#define ES_G (ES_EI + ES_NL)       // [32] landmark gradients
#define ES_N (ES_G + ES_NL + 8)    // 2712
#define E_X (E_WS + 2 * ES_N)    // [XSTRIDE] the window's state, fetched when the kernel starts (the candidate is formed from it at the end)
#define E_TOTAL (E_X + ((XSTRIDE + 7) & ~7))
static_assert(E_TOTAL * 8 <= 160 * 1024, "one workgroup's LDS");
static_assert(3920 <= 2 * 2560, "the pose system (pm_at) fits the hand-over buffers");
extern __shared__ __attribute__((aligned(16))) double e_lds[];   // the workgroup's dynamic LDS (k_solve_mw8 and its roles)
// Its address, once per role. In a function that is not a kernel the address of dynamic LDS is a scalar load from a per-kernel table, and
// the compiler repeats that load wherever it needs the address rather than keep a register: every such load is an `s_waitcnt lgkmcnt(0)`
// in the middle of the LDS operand stream (scalar loads return out of order, so the counter cannot be waited on partially) — the matrix
// waves waited for the next half trip's operands before every half trip. The empty asm makes the value opaque, so it stays in its SGPR.
typedef __attribute__((address_space(3))) double lds_double;
__device__ __forceinline__ double *e_lds_base() {
  unsigned off = (unsigned)(unsigned long long)(lds_double *)e_lds;
  off = __builtin_amdgcn_readfirstlane(off);
  asm volatile("" : "+s"(off));
  return (double *)(lds_double *)(unsigned long long)off;
}
// The same LDS pointer held in a vector register (opaque to the compiler): for tables read at many compile-time offsets with the same
// address in every lane. From a scalar base the compiler forms every address on the scalar unit (one SGPR each, hoisted out of the loops,
// then spilled to VGPR lanes: v_readlane + v_mov in front of every read); from a vector base they are immediate offsets of one register.
__device__ __forceinline__ double *lds_in_vgpr(double *p) {
  unsigned off = (unsigned)(unsigned long long)(lds_double *)p;
  asm volatile("" : "+v"(off));
  return (double *)(lds_double *)(unsigned long long)off;
}

// Every role below is a function of its own (__noinline__): the kernel's eight waves share one register allocation of 256 per lane, and
// with the roles inlined into one body the allocator spilled all over it (each role's live ranges crowd the others'; measured: 100 - 500
// spilled registers, the loads of the hot loops waiting in scratch). A call costs the callee-saved registers' round trip, once per role
// and linearisation. The roles find the workgroup's LDS through the dynamic-LDS symbol; their inputs come as (wave-uniform) arguments.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ double uni(double x) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x))); }
// a role's pointer arguments, wave-uniform and typed as what they are — global memory. (Through a generic pointer a role that is not a
// kernel loads with flat instructions; those count on both memory counters and return out of order, so while one may be in flight the
// compiler waits for lgkmcnt(0) before anything that depends on LDS: no LDS prefetch survives.)
#define GLOBAL_AS __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ T *uni(T *p);
template <class T> __device__ __forceinline__ GLOBAL_AS T *uni_g(T *p) { return (GLOBAL_AS T *)uni(p); }
template <class T> __device__ __forceinline__ T *uni(T *p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}
// main-loop geometry shared by the roles: chain step c at iteration c, its T(k) at iteration c + 1, the rank update at iteration c + 2;
// enough iterations for the landmark Schur complement to pass one LDS slice (four half trips) per iteration
struct E_Geom { int mid, nL, nU, nS, NH, NT, NSTEP; };
__device__ __forceinline__ E_Geom e_geom(int F, int L) {
  E_Geom G;
  G.mid = (F - 1) >> 1; G.nL = G.mid; G.nU = F - 1 - G.mid; G.nS = G.nU;
  const int nks = (L + 3) >> 2;
  G.NT = (nks + 3) >> 2; G.NH = (nks + 1) >> 1;
  G.NSTEP = max(G.nS + 3, (G.NH + 3) >> 2);
  return G;
}
__device__ __forceinline__ int e_slice_h0(const E_Geom &G, int i) { return (i * G.NH) / G.NSTEP; }   // slice i = half trips [h0(i), h0(i + 1)): at most four
// The matrix waves' operands reach them through LDS, a slice of 32 landmarks (all 80 coupling rows, 1 / (E + mu dhat^2), the gradients)
// per iteration of the main loop: the chain and T waves — the four that have a few registers to spare — fetch the next iteration's
// slice at the start of an iteration (coalesced, 11 loads per lane, in flight behind the iteration's own work) and store it before the
// iteration's barrier. With the operands straight from global memory, as the forms with one or two matrix waves have them, a half trip's
// ~1 k cycles of matrix instructions wait for a ~3 k cycle round trip on a nearly idle chip.
#define ES_NLD 11   // loads per lane of a loader wave (4 x 64 x 11 >= 80 x 32 + 64)
struct E_Slice {
  double r[ES_NLD];
  template <class P>
  __device__ __forceinline__ void load(const E_Geom &G, int i, int t256, P wl, P lm_einv, P lm_g, int L) {
    const int l0 = 8 * e_slice_h0(G, i);
#pragma unroll
    for (int u = 0; u < ES_NLD; ++u) {
      const int e = t256 + 256 * u;
      double val = 0.0;
      if (e < 80 * ES_NL) { const int a = e >> 5, l = l0 + (e & 31); if (l < L) val = wl[(size_t)a * L + l]; }
      else if (e < 80 * ES_NL + ES_NL) { const int l = l0 + (e - 80 * ES_NL); if (l < L) val = lm_einv[l]; }
      else if (e < 80 * ES_NL + 2 * ES_NL) { const int l = l0 + (e - 80 * ES_NL - ES_NL); if (l < L) val = lm_g[l]; }
      r[u] = val;
    }
  }
  __device__ __forceinline__ void store(double *lds, int i, int t256) const {
    double *Ws = lds + E_WS + ES_N * (i & 1);
#pragma unroll
    for (int u = 0; u < ES_NLD; ++u) {
      const int e = t256 + 256 * u;
      if (e < 80 * ES_NL) Ws[(e >> 5) * ES_LD + (e & 31)] = r[u];
      else if (e < 80 * ES_NL + ES_NL) Ws[ES_EI + (e - 80 * ES_NL)] = r[u];
      else if (e < 80 * ES_NL + 2 * ES_NL) Ws[ES_G + (e - 80 * ES_NL - ES_NL)] = r[u];
    }
  }
};

// chain scratch (per chain wave)
#define MC_LM 0
#define MC_TA0 176
#define MC_SN 528
// Cholesky (E_CHOL region)
#define MX_D16 0
#define MX_LI16 272
#define MX_P16 560
#define MX_PANEL 1024
// reduction slots (per wave: 8 each)
#define QR_GN 0
#define QR_GMAX 8
#define QR_Q 16
#define QR_GNN 24
#define QR_GY 32
#define QR_QX 40
#define QR_FAIL 48     // [4] chain down, chain up, pose system
#define QR_CA 52
#define QR_CB 53
#define QR_GO 54

extern "C" size_t vilo_solve_mw8_lds_bytes() { return (size_t)E_TOTAL * sizeof(double); }

__device__ __forceinline__ void e_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void e_barrier_global() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#define E_BARRIER() e_barrier()
#define E_BARRIER_GLOBAL() e_barrier_global()
// (profiling build: cycles a role waits at the main loop's step barriers, left in red[QR_WAIT + slot] for the kernel body)
#define E_STEP_BARRIER() do { const long long w0_ = pclk64(); e_barrier(); PCLK(ewait += clock64() - w0_); } while (0)
#define QR_WAIT 56
// tiles of the pose system owned by the matrix waves (bit t of the mask = tile t of c_tI / c_tJ) and the blocks X of the reduced
// right-hand side each of them accumulates (a wave has the operands of the blocks its tiles touch):
//   B1: (0,0) (1,0) (1,1) (2,0)   rhs 0, 1        B2: (2,1) (2,2) (3,0) (3,1)   rhs 2
//   B3: (3,2) (3,3) (4,0) (4,1)   rhs 3           B4: (4,2) (4,3) (4,4)         rhs 4
#define EB1_MASK 0x000f
#define EB2_MASK 0x00f0
#define EB3_MASK 0x0f00
#define EB4_MASK 0x7000
constexpr int e_blocks_of(int mask, int rhs) {   // operand blocks a matrix wave needs: those of its tiles' rows and columns + its rhs blocks
  constexpr int tI[15] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4}, tJ[15] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4};
  int m = rhs;
  for (int t = 0; t < 15; ++t)
    if ((mask >> t) & 1) m |= (1 << tI[t]) | (1 << tJ[t]);
  return m;
}

// The pose system in LDS while it is factored (over the T hand-over buffers): the lower block triangle, row-major — row i of block row
// I = i / 16 holds the columns 0 .. 16 I + 15 with the odd stride 16 (I + 1) + 1, so a lane that owns a row, a lane that owns a column and a
// matrix-instruction operand read (16 rows x 4 columns) all reach distinct banks. 3920 doubles.
__device__ __forceinline__ int pm_at(int i, int c) {
  const int I = i >> 4;
  return E_TILES + 128 * I * (I + 1) + 16 * I + (i & 15) * (16 * (I + 1) + 1) + c;
}
#define E_RINV E_CHOL            // [80] 1 / L_cc
#define E_LD16 (E_CHOL + 80)     // [16][17] the diagonal tile's factor L_jj of the block column at hand (fixed place: compile-time addresses)
#define E_RV16 (E_CHOL + 352)    // [16] its 1 / L_cc

// The 80 x 80 pose system's Cholesky, forward and backward solve by all eight waves. In: the matrix (pm_at) without its mu D^2, dh2, the
// reduced right-hand side v. Out: y (E_Y), masked to the active dimensions. Returns 1 (on the factor wave, cw 0) if a pivot failed.
//
// Right-looking by block columns of 16. The critical path is one wave's (cw 0): factor the diagonal tile (a lane per row, v_readlane
// broadcasts: ~5 k cycles), solve the next block row against it (a lane per row again: x L_jj^T = a with L's entries as broadcast LDS
// reads, no inverse is ever formed), take that block's update of the next diagonal tile on the matrix cores, factor again. The right-hand
// side rides in lane 16 of the same solve. Everything else happens beside it: cw 1 solves the other block rows of the panel (up to 48 rows
// at once), cw 2 .. 7 apply the panel to the trailing tiles (4 matrix instructions each) while cw 0 is already factoring the next
// diagonal tile. Two workgroup barriers per block column: L_jj is there / the panel is there. The backward solve L^T x = y is 80 steps
// of a column sweep on cw 0 (a lane per unknown, the row of L a coalesced LDS read issued ahead).
// The one-wave version this replaces (factor + inverse of each diagonal tile, panels, trailing updates and both solves on B1 with the
// other seven waves waiting) took 69 k of the solve's 198 k cycles.
__device__ __noinline__ int mw8_chol80(int cw_, double mu, int F, int cmask) {
  double *const lds = e_lds_base();
  const int cw = uni(cw_);
  mu = uni(mu); F = uni(F); cmask = uni(cmask);
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  double *dh2 = lds + E_DH2, *y = lds + E_Y, *v = lds + E_V, *rinvs = lds + E_RINV;
  double *const LD16 = lds_in_vgpr(lds + E_LD16), *const RV16 = LD16 + (E_RV16 - E_LD16);
  int fail = 0;
  const long long cclk0 = pclk64();
#define CSTAMP(n_) PCLK(if (cw == 0 && lane == 0) lds[E_RED + 76 + (n_)] = (double)(clock64() - cclk0))
  // x L_jj^T = a for the rows the lanes hold (x: a on entry), column by column: x_q = x_q / L_qq, then x_c -= x_q L_cq for c > q. L's
  // entries are broadcast LDS reads (the same address in every lane); the reads of a column are issued two columns ahead of its use and
  // pinned there — left to itself the compiler keeps two reads in flight and the solve waits ~70 times for an LDS round trip (9.5 k
  // cycles for what is 136 multiply-adds per lane).
  auto row_solve = [&](double (&x)[16]) {
    double Lq[16][16], rv[16];
    auto ldcol = [&](int q) {
      rv[q] = RV16[q];
#pragma unroll
      for (int c = q + 1; c < 16; ++c) Lq[q][c] = LD16[c * 17 + q];
    };
    ldcol(0); ldcol(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q + 2 < 16) ldcol(q + 2);
      __builtin_amdgcn_sched_barrier(0);
      x[q] *= rv[q];
#pragma unroll
      for (int c = q + 1; c < 16; ++c) x[c] -= x[q] * Lq[q][c];
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // tile (I, J) -= P_I P_J^T, P the panel of block column j
  auto trailing = [&](int I, int J, int j) {
    mfma_d4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = lds[pm_at(16 * I + lk + 4 * r, 16 * J + lr)];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-lds[pm_at(16 * I + lr, 16 * j + 4 * kk + lk)], lds[pm_at(16 * J + lr, 16 * j + 4 * kk + lk)], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[pm_at(16 * I + lk + 4 * r, 16 * J + lr)] = acc[r];
  };
#pragma unroll 1
  for (int j = 0; j < 5; ++j) {
    if (cw == 0) {
      // the diagonal tile, a lane per row (the four lane groups do the same work), + mu D^2
      double a[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = lds[pm_at(16 * j + lr, 16 * j + c)];
      {
        const double md = mu * dh2[16 * j + lr];
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] += (c == lr) ? md : 0.0;
      }
      double myrinv = 1.0;
      fail |= chol_rows<16>(a, lr, myrinv);
      if (lk == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { lds[pm_at(16 * j + lr, 16 * j + c)] = a[c]; LD16[lr * 17 + c] = a[c]; }   // (zeros above the diagonal)
        rinvs[16 * j + lr] = myrinv; RV16[lr] = myrinv;
      }
    }
    CSTAMP(2 * j);
    E_BARRIER();   // L_jj
    double x[16];
    const int i1 = 16 * (j + 2) + lane;   // cw 1's row
    const bool isr1 = cw == 1 && lane < 16 * (3 - j);
    if (cw == 0) {
      if (j < 4) {
        // block row j + 1 (every lane group solves the same 16 rows: the matrix instructions below take their operands from registers)
        const int ib = 16 * (j + 1) + lr;
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = lds[pm_at(ib, 16 * j + c)];
        mfma_d4 acc;   // the next diagonal tile
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = lds[pm_at(16 * (j + 1) + lk + 4 * r, 16 * (j + 1) + lr)];
        row_solve(x);
        if (lk == 0) {
#pragma unroll
          for (int c = 0; c < 16; ++c) lds[pm_at(ib, 16 * j + c)] = x[c];
        }
        // D_{j+1} -= P P^T ahead of everything else: lane (lr, lk) supplies P[lr][4 kk + lk] as both operands
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const double p = lk == 0 ? x[4 * kk] : (lk == 1 ? x[4 * kk + 1] : (lk == 2 ? x[4 * kk + 2] : x[4 * kk + 3]));
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-p, p, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[pm_at(16 * (j + 1) + lk + 4 * r, 16 * (j + 1) + lr)] = acc[r];
      }
    } else if (cw == 1) {
      // the other block rows of the panel (up to 48 rows), the right-hand side in lane 63
      const bool isv = lane == 63;
#pragma unroll
      for (int c = 0; c < 16; ++c) {   // (both reads unconditional: every address exists)
        const double tr = lds[pm_at(min(i1, 79), 16 * j + c)], tv = v[16 * j + c];
        x[c] = isr1 ? tr : (isv ? tv : 0.0);
      }
      row_solve(x);
      if (isr1) {
#pragma unroll
        for (int c = 0; c < 16; ++c) lds[pm_at(i1, 16 * j + c)] = x[c];
      }
      if (lane == 63) {   // y_j (the backward solve reads it after the last of these barriers)
#pragma unroll
        for (int c = 0; c < 16; ++c) y[16 * j + c] = x[c];
      }
    }
    CSTAMP(2 * j + 1);
    E_BARRIER();   // the panel, y_j
    if (cw == 1) {
      // the right-hand side of the rows below loses P y_j (after the barrier: the factor wave is not kept waiting for it)
      double yj[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) yj[c] = readlane_d(x[c], 63);
      if (j < 4) {
        double s = 0.0, s1 = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) { s += x[c] * yj[c]; s1 += lds[pm_at(16 * (j + 1) + lr, 16 * j + c)] * yj[c]; }   // (block row j + 1's panel rows are cw 0's)
        if (isr1) v[i1] -= s;
        if (lane < 16) v[16 * (j + 1) + lr] -= s1;
      }
    }
    if (cw >= 2) {
      int n = 0;
#pragma unroll 1
      for (int I = j + 1; I < 5; ++I)
#pragma unroll 1
        for (int J = j + 1; J <= I; ++J) {
          if (I == j + 1 && J == j + 1) continue;   // (cw 0's)
          if (cw - 2 == n % 6) trailing(I, J, j);
          ++n;
        }
    }
  }
  // L^T x = y on cw 0: from the last unknown up, a lane per unknown (lane c & 63; z0: 0 .. 63, z1: 64 .. 79). Step c: x_c = y_c / L_cc, then
  // y_q -= L_cq x_c for q < c. The lanes carry z_q = y_q / L_qq instead of y_q (the row of L is scaled by 1 / L_qq when it arrives, off the
  // dependent chain), so a step is v_readlane -> multiply-add; x_c goes to its lane of xs by v_writelane, and the lanes at and beyond c,
  // which are never read again, may take whatever the row's read brings (no masks). Rows are read four steps ahead.
  if (cw == 0) {
    lds_fence();
    const double r0 = rinvs[lane], r1 = (lane < 16) ? rinvs[64 + lane] : 0.0;
    double z0 = y[lane] * r0, z1 = (lane < 16) ? y[64 + lane] * r1 : 0.0;
    int xs0h = 0, xs0l = 0, xs1h = 0, xs1l = 0;
    const double *rowp = lds + lane;
    double l0v[80], l1v[80];
    auto ldrow = [&](int c) {
      if (c > 0) l0v[c] = rowp[pm_at(c, 0)];
      if (c > 64) l1v[c] = rowp[pm_at(c, 64)];   // (lanes 0 .. c - 65; the others read on into LDS that exists)
    };
    ldrow(79); ldrow(78); ldrow(77); ldrow(76);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 79; c >= 0; --c) {
      if (c >= 4) ldrow(c - 4);
      __builtin_amdgcn_sched_barrier(0);
      const double zsrc = (c >= 64) ? z1 : z0;
      const int xl = __builtin_amdgcn_readlane(__double2loint(zsrc), c & 63), xh = __builtin_amdgcn_readlane(__double2hiint(zsrc), c & 63);
      const double xc = __hiloint2double(xh, xl);
      if (c >= 64) { asm("v_writelane_b32 %0, %1, %2" : "+v"(xs1l) : "s"(xl), "i"(c & 63)); asm("v_writelane_b32 %0, %1, %2" : "+v"(xs1h) : "s"(xh), "i"(c & 63)); }
      else { asm("v_writelane_b32 %0, %1, %2" : "+v"(xs0l) : "s"(xl), "i"(c & 63)); asm("v_writelane_b32 %0, %1, %2" : "+v"(xs0h) : "s"(xh), "i"(c & 63)); }
      if (c > 64) z1 -= (l1v[c] * r1) * xc;
      if (c > 0) z0 -= (l0v[c] * r0) * xc;
      __builtin_amdgcn_sched_barrier(0);
    }
    const double y0 = __hiloint2double(xs0h, xs0l), y1 = __hiloint2double(xs1h, xs1l);
    y[lane] = cd_active(lane, F, cmask) ? y0 : 0.0;
    if (lane < 16) y[64 + lane] = cd_active(64 + lane, F, cmask) ? y1 : 0.0;
  }
  CSTAMP(10);
  return fail;
}

// waves C1 / C2: the twisted chain, 13 x 13 part (see the head of the file). The failure flag goes to red[QR_FAIL + (up ? 1 : 0)].
__device__ __noinline__ void mw8_role_chain(int up_, const double *bimg_, const double *wl_, const double *lm_einv_, const double *lm_g_, double mu, int F, int L) {
  double *const lds = e_lds_base();
  const bool up = uni(up_) != 0;   // C2 walks the frames upwards from 0
  const auto bimg = uni_g(bimg_), wl = uni_g(wl_), lm_einv = uni_g(lm_einv_), lm_g = uni_g(lm_g_);
  mu = uni(mu); F = uni(F); L = uni(L);
  const E_Geom G = e_geom(F, L);
  const int mid = G.mid, nL = G.nL, nU = G.nU, nS = G.nS, NSTEP = G.NSTEP;
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  double *red = lds + E_RED, *DB = lds + E_DB;
  int fail = 0;
  long long ewait = 0, fclk0 = 0;
  const int t256 = (up ? 64 : 0) + lane;   // loader index (C1, C2, TD, TU)
  E_Slice sl;
  sl.load(G, 0, t256, wl, lm_einv, lm_g, L);
  sl.store(lds, 0, t256);
  const int grp = lk, c = lr;
  const int row = c < 13 ? c : 0;
  double *scr = lds + (up ? E_CH2 : E_CH1);
  double *LM = scr + MC_LM, *SN = scr + MC_SN, *TAcur = scr + MC_TA0;
  // one frame of a chain. nb: the neighbour this frame hands its Schur update to (-1: none: the middle frame); prevk: the frame
  // eliminated before this one in the same direction (-1: first)
  PCLK(fclk0 = 0);
  auto frame = [&](int k, int nb, int prevk, bool middle) {
#define FSTAMP(n_) PCLK(if (!up && prevk >= 0 && !middle && k == F - 2 && lane == 0) red[64 + (n_)] = (double)(clock64() - fclk0))
    PCLK(fclk0 = clock64());
    double a[13], l[13], rhs[13], adn[4];
    // right-hand sides of the off-diagonal factor: column `row` of A_{k,k-1} (downwards), of A_{k+1,k}^T (upwards)
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      double rv = 0.0;
      if (nb >= 0) rv = up ? bimg[BI_AOT + (k * 13 + i) * 13 + row] : bimg[BI_AOT + ((k - 1) * 13 + row) * 13 + i];
      rhs[i] = rv;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) adn[r] = (nb >= 0 && lr < 13 && lk + 4 * r < 13) ? bimg[BI_AD + nb * 169 + (lk + 4 * r) * 13 + lr] : 0.0;   // A_{nb,nb}, accumulator order
    // S_k (lane = row): the first frame of a direction straight from A_kk, later ones from the update left by the previous step
    if (prevk < 0) {
#pragma unroll
      for (int j = 0; j < 13; ++j) a[j] = bimg[BI_AD + (k * 13 + row) * 13 + j];
    } else {
#pragma unroll
      for (int j = 0; j < 13; ++j) a[j] = SN[row * 13 + j];
    }
    if (middle && nL > 0) {
      const double *XS = lds + E_XSN;
#pragma unroll
      for (int j = 0; j < 13; ++j) a[j] -= XS[row * 13 + j];
    }
    {
      const double md = mu * DB[13 * k + row];
#pragma unroll
      for (int j = 0; j < 13; ++j) a[j] += (j == row) ? md : 0.0;
    }
    FSTAMP(0);
    double myrinv = 1.0;
    fail |= chol_rows<13>(a, c, myrinv);
#pragma unroll
    for (int j = 0; j < 13; ++j) l[j] = a[j];
#pragma unroll
    for (int j = 0; j < 13; ++j) asm volatile("" : "+v"(l[j]));
    FSTAMP(1);
    double cl[13];
#pragma unroll
    for (int q2 = 0; q2 < 13; ++q2) {
      double vv = (grp == 0) ? rhs[q2] : ((q2 == c) ? 1.0 : 0.0);
#pragma unroll
      for (int q = 0; q < q2; ++q) vv -= readlane_d(l[q], q2) * cl[q];
      cl[q2] = vv * readlane_d(myrinv, q2);
      asm volatile("" : "+v"(cl[q2]));
      __builtin_amdgcn_sched_barrier(0);
    }
    FSTAMP(2);
    double *Mk = lds + E_M + 176 * k, *TAk = lds + E_TA + 176 * k;
    if (c < 13 && grp < 2) {
      if (grp == 0) {
#pragma unroll
        for (int q = 0; q < 13; ++q) { TAcur[q * 13 + c] = cl[q]; TAk[q * 13 + c] = cl[q]; }
      } else {
#pragma unroll
        for (int q = 0; q < 13; ++q) { LM[q * 13 + c] = cl[q]; Mk[q * 13 + c] = cl[q]; }
      }
    }
    lds_fence();
    FSTAMP(3);
    // the neighbour's diagonal block loses T_A^T T_A: one 16 x 16 tile on the matrix cores. The frame next to the middle leaves C2's
    // share where C1 finds it (C1's own share goes into its SN with A_mm)
    if (nb >= 0) {
      mfma_d4 sn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int q = 4 * kk + lk;
        const double ta = ((lr < 13) && (q < 13)) ? TAcur[min(q, 12) * 13 + min(lr, 12)] : 0.0;
        sn = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, ta, sn, 0, 0, 0);
      }
      const bool to_mid = up && nb == mid;
      double *dst = to_mid ? lds + E_XSN : SN;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lr < 13 && lk + 4 * r < 13) dst[(lk + 4 * r) * 13 + lr] = to_mid ? sn[r] : adn[r] - sn[r];
    }
    FSTAMP(4);
#undef FSTAMP
  };
  E_BARRIER();   // (the matrix waves' skip table and slice 0: every wave meets the same barriers)
  for (int i = 0; i < NSTEP; ++i) {
    if (i + 1 < NSTEP) sl.load(G, i + 1, t256, wl, lm_einv, lm_g, L);
    if (!up) {
      if (i < nU) { const int k = F - 1 - i; frame(k, k - 1, i == 0 ? -1 : k + 1, false); }
      else if (i == nS) frame(mid, -1, mid + 1, true);
    } else if (i < nL) {
      frame(i, i + 1, i == 0 ? -1 : i - 1, false);
    }
    if (i == nS && lane == 0) red[QR_FAIL + (up ? 1 : 0)] = (double)fail;
    if (i + 1 < NSTEP) sl.store(lds, i + 1, t256);
    E_STEP_BARRIER();
  }
  PCLK(if (lane == 0) red[QR_WAIT + (up ? 1 : 0)] = (double)ewait);
  E_BARRIER();   // the tiles and the reduced right-hand side
}

// waves TD / TU: the chain's coupling rows T(k), one step behind their chain wave (and a quarter of the operand slices: E_Slice).
__device__ __noinline__ void mw8_role_T(int up_, const double *bimg_, const double *wl_, const double *lm_einv_, const double *lm_g_, int F, int L, int kb) {
  double *const lds = e_lds_base();
  const bool up = uni(up_) != 0;
  const auto bimg = uni_g(bimg_), wl = uni_g(wl_), lm_einv = uni_g(lm_einv_), lm_g = uni_g(lm_g_);
  F = uni(F); L = uni(L); kb = uni(kb);
  const E_Geom G = e_geom(F, L);
  const int mid = G.mid, nL = G.nL, nU = G.nU, nS = G.nS, NSTEP = G.NSTEP;
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  double *GB = lds + E_GB;
  const int t256 = 128 + (up ? 64 : 0) + lane;   // loader index (C1, C2, TD, TU)
  long long ewait = 0;
  E_Slice sl;
  sl.load(G, 0, t256, wl, lm_einv, lm_g, L);
  sl.store(lds, 0, t256);
  int fX[5], oX[5];
#pragma unroll
  for (int X = 0; X < 5; ++X) { const int col = 16 * X + lr; fX[X] = col < 66 ? col / 6 : 99; oX[X] = col < 66 ? col - 6 * fX[X] : 0; }
  double *Tbase = lds + (up ? E_T2 : E_T1);
  mfma_d4 T[5];
#pragma unroll
  for (int X = 0; X < 5; ++X) T[X] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  // T(k) of the frame the chain wave of this direction eliminated in step `step` (prevk: the frame before it in that direction, -1: first)
  auto tframe = [&](int k, int prevk, bool middle, int step) {
    const int x_lo = (up || middle || k <= kb) ? 0 : max(0, (6 * (k - 1)) >> 4);   // (upwards and in the middle T(k) is dense)
    mfma_d4 V[5];
    // [B_k | g_k] in accumulator order: row lk + 4 r, column 16 X + lr; column 79 carries the gradient
#pragma unroll
    for (int X = 0; X < 5; ++X) {
      const int df = fX[X] - k + 1;
      const bool on = df >= 0 && df <= 2;
      const auto *src = bimg + BI_BS + (k * 16 + lk) * 18 + 6 * min(max(df, 0), 2) + oX[X];
#pragma unroll
      for (int r = 0; r < 4; ++r) V[X][r] = on ? src[72 * r] : 0.0;
    }
    if (k == kb) {
#pragma unroll
      for (int X = 0; X < 5; ++X)
#pragma unroll
        for (int r = 0; r < 4; ++r) V[X][r] += bimg[BI_BP + (lk + 4 * r) * 80 + 16 * X + lr];
    }
    if (lr == 15) {
#pragma unroll
      for (int r = 0; r < 4; ++r) V[4][r] = (lk + 4 * r < 13) ? GB[13 * k + lk + 4 * r] : 0.0;
    }
    // V -= T_A(prev)^T T(prev);  T(k) = M_k V
    const double *TAprev = lds + E_TA + 176 * max(prevk, 0), *LM = lds + E_M + 176 * k;
    double at[4], am[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int q = 4 * kk + lk;
      const bool in = (lr < 13) && (q < 13);
      const double ta = TAprev[min(q, 12) * 13 + min(lr, 12)], m = LM[min(lr, 12) * 13 + min(q, 12)];
      at[kk] = (in && prevk >= 0) ? -ta : 0.0;
      am[kk] = in ? m : 0.0;
    }
    if (prevk >= 0) {
#pragma unroll
      for (int X = 0; X < 5; ++X)
        if (X >= x_lo) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) V[X] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[kk], T[X][kk], V[X], 0, 0, 0);
        }
    }
    if (middle && nL > 0) {
      // the lower neighbour's term: T'_A(m-1) from the factor store, T(m-1) from TU's hand-over buffer (its last frame)
      const double *TAl = lds + E_TA + 176 * (mid - 1);
      const double *Tl = lds + E_T2 + 1280 * ((nL - 1) & 1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int q = 4 * kk + lk;
        const double ta = ((lr < 13) && (q < 13)) ? -TAl[min(q, 12) * 13 + min(lr, 12)] : 0.0;
#pragma unroll
        for (int X = 0; X < 5; ++X) V[X] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, Tl[(X * 4 + kk) * 64 + lane], V[X], 0, 0, 0);
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
    // hand T(k) over (tiles left of x_lo are zero: written as such, the consumers need no per-frame sparsity table)
    double *Tb = Tbase + 1280 * (step & 1);
#pragma unroll
    for (int X = 0; X < 5; ++X)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) Tb[(X * 4 + kk) * 64 + lane] = (X >= x_lo) ? T[X][kk] : 0.0;
  };
  E_BARRIER();   // (skip table, slice 0)
  for (int i = 0; i < NSTEP; ++i) {
    if (i + 1 < NSTEP) sl.load(G, i + 1, t256, wl, lm_einv, lm_g, L);
    const int cstep = i - 1;   // the chain step whose M_k / T_A(k) are in LDS now
    if (cstep >= 0) {
      if (!up) {
        if (cstep < nU) { const int k = F - 1 - cstep; tframe(k, cstep == 0 ? -1 : k + 1, false, cstep); }
        else if (cstep == nS) tframe(mid, mid + 1, true, cstep);
      } else if (cstep < nL) {
        tframe(cstep, cstep == 0 ? -1 : cstep - 1, false, cstep);
      }
    }
    if (i + 1 < NSTEP) sl.store(lds, i + 1, t256);
    E_STEP_BARRIER();
  }
  PCLK(if (lane == 0) lds[E_RED + QR_WAIT + 2 + (up ? 1 : 0)] = (double)ewait);
  E_BARRIER();   // the tiles and the reduced right-hand side
}

// waves B1 .. B4: the pose system's tiles MASK — landmark Schur complement from the LDS slices, rank updates from the T waves' hand-over
// buffers, the blocks RHS of the reduced right-hand side; at the end the tiles go to E_TILES for the Cholesky.
template <int MASK, int RHS>
__device__ __noinline__ void mw8_role_B(int bw_, const double *Cimg_, const unsigned char *lms_, int F, int L) {
  double *const lds = e_lds_base();
  const int bw = uni(bw_);
  const auto Cimg = uni_g(Cimg_);
  const auto lms = uni_g(lms_);
  F = uni(F); L = uni(L);
  const E_Geom G = e_geom(F, L);
  const int nL = G.nL, nU = G.nU, nS = G.nS, NSTEP = G.NSTEP, NT = G.NT;
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  double *g = lds + E_G, *v = lds + E_V;
  constexpr int NEED = e_blocks_of(MASK, RHS);            // operand blocks the wave reads
  long long ewait = 0;
  mfma_d4 acc[15];   // (only the tiles of MASK exist)
#pragma unroll
  for (int t = 0; t < 15; ++t)
    if ((MASK >> t) & 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = Cimg[(t * 4 + r) * 64 + lane];
    }
  double yacc[5], yr[5];
#pragma unroll
  for (int X = 0; X < 5; ++X) { yacc[X] = 0.0; yr[X] = 0.0; }
  // trips (16 landmarks) whose landmarks all start at frame 3 or later have no coupling to the first 16 pose columns: bit tr of a wave-uniform
  // mask (the landmarks are sorted by start frame; trips beyond the mask's 64 are simply not skipped)
  unsigned long long skipmask = 0;
  if (L > 0) skipmask = __ballot(lane < NT + 2 && 6 * (int)lms[min(16 * lane, L - 1)] >= 16);
  auto skip_of = [&](int h) -> bool { const int tr = h >> 1; return tr < 64 && ((skipmask >> tr) & 1); };
  E_BARRIER();   // (slice 0 is the chain and T waves'; every wave meets this barrier)
  // operands of a half trip (two k-steps of 4 landmarks) from the iteration's slice, the next half's behind this one's matrix instructions
  // (hl: the half's position inside the slice; the two buffers and the slice's at most four halves are unrolled: every operand and every
  // accumulator keeps its registers — with the buffer picked at run time the compiler moved the accumulators after every half trip and
  // waited for the matrix pipe to drain first: 1650 cycles per half trip of 8 matrix instructions)
  double opb[2][2][5], eb[2][2], gb[2][2];
  auto ldhalf = [&](int hl, const double *Ws, auto bs) {
    constexpr int bsel = decltype(bs)::value;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int lc = 8 * hl + 4 * u + lk;
      eb[bsel][u] = Ws[ES_EI + lc]; gb[bsel][u] = Ws[ES_G + lc];
#pragma unroll
      for (int X = 0; X < 5; ++X)   // (block 0 of a skipped trip too)
        if ((NEED >> X) & 1) opb[bsel][u][X] = Ws[(16 * X + lr) * ES_LD + lc];
    }
  };
  auto dohalf = [&](bool sk, auto bs) {
    constexpr int bsel = decltype(bs)::value;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const double ei = eb[bsel][u], ge = gb[bsel][u] * ei;
#pragma unroll
      for (int t = 0; t < 15; ++t)
        if (((MASK >> t) & 1) && c_tJ[t] > 0)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-(opb[bsel][u][c_tI[t]] * ei), opb[bsel][u][c_tJ[t]], acc[t], 0, 0, 0);
#pragma unroll
      for (int X = 1; X < 5; ++X)
        if ((RHS >> X) & 1) yacc[X] += opb[bsel][u][X] * ge;
    }
    if (!sk) {   // the tiles of the first 16 columns
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const double ei = eb[bsel][u];
#pragma unroll
        for (int t = 0; t < 15; ++t)
          if (((MASK >> t) & 1) && c_tJ[t] == 0)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-(opb[bsel][u][c_tI[t]] * ei), opb[bsel][u][0], acc[t], 0, 0, 0);
        if (RHS & 1) yacc[0] += opb[bsel][u][0] * (gb[bsel][u] * ei);
      }
    }
  };
    // C -= T^T T for one delivered frame (all five tile columns: the T waves write zeros where T(k) has none)
    auto rank_update = [&](const double *Tb) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double Tk[5];
#pragma unroll
        for (int X = 0; X < 5; ++X) Tk[X] = ((NEED >> X) & 1 || X == 4) ? Tb[(X * 4 + kk) * 64 + lane] : 0.0;
        const double tg = __shfl(Tk[4], (lane & 48) | 15, 64);   // t_g(k) (column 79): the pose system must not see it
        if (lr == 15) Tk[4] = 0.0;
#pragma unroll
        for (int t = 0; t < 15; ++t)
          if ((MASK >> t) & 1) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Tk[c_tI[t]], Tk[c_tJ[t]], acc[t], 0, 0, 0);
#pragma unroll
        for (int X = 0; X < 5; ++X)
          if ((RHS >> X) & 1) yr[X] += Tk[X] * tg;
      }
    };
  for (int i = 0; i < NSTEP; ++i) {
    const int h0 = e_slice_h0(G, i), h1 = e_slice_h0(G, i + 1);
    const double *Ws = lds + E_WS + ES_N * (i & 1);
    const long long bclk0 = pclk64();
#define BSTAMP(n_) PCLK(if (bw == 2 && i == 3 && lane == 0) lds[E_RED + 72 + (n_)] = (double)(clock64() - bclk0))
    const int nh = h1 - h0;   // <= 4 (e_geom)
    // (the loads are unconditional — a slice always holds four halves, zero beyond the window's landmarks: a load under a branch costs
    //  the compiler its count of the loads in flight, and every half trip then waits for the next one's operands)
    ldhalf(0, Ws, std::integral_constant<int, 0>{});
    auto hstep = [&](auto hlc) {
      constexpr int hl = decltype(hlc)::value;
      if constexpr (hl < 3) ldhalf(hl + 1, Ws, std::integral_constant<int, (hl + 1) & 1>{});
      if (hl < nh) dohalf(skip_of(h0 + hl), std::integral_constant<int, hl & 1>{});
    };
    hstep(std::integral_constant<int, 0>{}); hstep(std::integral_constant<int, 1>{});
    hstep(std::integral_constant<int, 2>{}); hstep(std::integral_constant<int, 3>{});
    BSTAMP(0);
    if (i >= 2) {
      // what the T waves finished in iteration i - 1 = the chain's step i - 2: TD a frame above the middle or (step nS) the middle, TU a frame below
      const int cstep = i - 2;
      if (cstep < nU || cstep == nS) rank_update(lds + E_T1 + 1280 * (cstep & 1));
      if (cstep < nL) rank_update(lds + E_T2 + 1280 * (cstep & 1));
    }
    BSTAMP(1);
#undef BSTAMP
    E_STEP_BARRIER();
  }
  PCLK(if (lane == 0) lds[E_RED + QR_WAIT + 4 + bw] = (double)ewait);
    // reduced right-hand side: g_P - sum_k T_B^T t_g - sum_l w_l g_l / (E_l + mu dhat_l^2), the blocks this wave owns
#pragma unroll
    for (int X = 0; X < 5; ++X)
      if ((RHS >> X) & 1) {
        double s = yr[X] + yacc[X];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lk == 0) v[16 * X + lr] = g[16 * X + lr] - s;
      }
    // the tiles move to the Cholesky's layout (pm_at; accumulator register r of lane (lr, lk) is row lk + 4 r, column lr of the tile)
#pragma unroll
    for (int t = 0; t < 15; ++t)
      if ((MASK >> t) & 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[pm_at(16 * c_tI[t] + lk + 4 * r, 16 * c_tJ[t] + lr)] = acc[t][r];
      }
  E_BARRIER();   // every wave's tiles and the reduced right-hand side are there
}

// C1 / C2 after y_P: back-substitution of the speed / leg-bias part (twisted sweeps). Partial sums of |D y|^2 and g^T y of the wave's frames
// go to red[QR_GNN / QR_GY + wave].
__device__ __noinline__ void mw8_role_sweeps(int up_, int wave_, const double *bimg_, int F, int L, int kb, int cmask) {
  double *const lds = e_lds_base();
  const bool up = uni(up_) != 0;
  const int wave = uni(wave_);
  const auto bimg = uni_g(bimg_);
  F = uni(F); L = uni(L); kb = uni(kb); cmask = uni(cmask);
  const E_Geom G = e_geom(F, L);
  const int mid = G.mid, nL = G.nL, nU = G.nU;
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  double *y = lds + E_Y, *red = lds + E_RED, *DB = lds + E_DB, *GB = lds + E_GB, *U = lds + E_U, *YB = lds + E_YB;
  double part_gnn = 0.0, part_gy = 0.0;
  const int k_lo = up ? 0 : mid, k_hi = up ? mid - 1 : F - 1;   // this wave's frames (C1 owns the middle)
  // c = g_B - B yP for this wave's frames: the IMU part of B_k spans poses k-1 .. k+1, the prior part frame kb only
  {
    double bsv[3][18];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int e = min(lane + 64 * m, 142), k = e / 13;
#pragma unroll
      for (int s2 = 0; s2 < 18; ++s2) bsv[m][s2] = bimg[BI_BS + (16 * k + (e - 13 * k)) * 18 + s2];
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int e = lane + 64 * m, k = min(e, 142) / 13;
      double sacc = GB[min(e, 143)];
#pragma unroll
      for (int s2 = 0; s2 < 18; ++s2) sacc -= bsv[m][s2] * y[min(max(6 * (k - 1) + s2, 0), 79)];
      if (e < 143 && k >= k_lo && k <= k_hi) U[e] = sacc;
    }
  }
  lds_fence();
  if (kb >= k_lo && kb <= k_hi) {
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
  const double *Mall = lds + E_M, *TAall = lds + E_TA;
  // forward sweeps, both halves at once:   down  u_k = M_k (c_k - T_A(k+1)^T u_{k+1})   k = F-1 .. m+1
  //                                        up    u_k = M_k (c_k - T'_A(k-1)^T u_{k-1})  k = 0 .. m-1
  double uprev = 0.0;
  if (!up) {
    for (int k = F - 1; k > mid; --k) {
      double s2 = U[13 * k + row];
      if (k < F - 1) {
        const double *TB = TAall + 176 * (k + 1);
#pragma unroll
        for (int q = 0; q < 13; ++q) s2 -= TB[q * 13 + row] * readlane_d(uprev, q);
      }
      const double *MB = Mall + 176 * k;
      double u = 0.0;
#pragma unroll
      for (int q = 0; q < 13; ++q) u += MB[row * 13 + q] * readlane_d(s2, q);
      if (lane < 13) U[13 * k + lane] = u;
      uprev = u;
    }
  } else {
    for (int k = 0; k < mid; ++k) {
      double s2 = U[13 * k + row];
      if (k > 0) {
        const double *TB = TAall + 176 * (k - 1);
#pragma unroll
        for (int q = 0; q < 13; ++q) s2 -= TB[q * 13 + row] * readlane_d(uprev, q);
      }
      const double *MB = Mall + 176 * k;
      double u = 0.0;
#pragma unroll
      for (int q = 0; q < 13; ++q) u += MB[row * 13 + q] * readlane_d(s2, q);
      if (lane < 13) U[13 * k + lane] = u;
      uprev = u;
    }
    if (lane < 13) lds[E_XU + lane] = uprev;   // u of frame m - 1 for the middle
  }
  E_BARRIER();
  // the middle frame (C1): u_m = M_m (c_m - T_A(m+1)^T u_{m+1} - T'_A(m-1)^T u_{m-1}),  y_m = M_m^T u_m
  double ymid = 0.0;
  if (!up) {
    double s2 = U[13 * mid + row];
    if (nU > 0) {
      const double *TB = TAall + 176 * (mid + 1);
#pragma unroll
      for (int q = 0; q < 13; ++q) s2 -= TB[q * 13 + row] * readlane_d(uprev, q);
    }
    if (nL > 0) {
      const double *TB = TAall + 176 * (mid - 1);
      const double ul = lds[E_XU + row];
#pragma unroll
      for (int q = 0; q < 13; ++q) s2 -= TB[q * 13 + row] * readlane_d(ul, q);
    }
    const double *MB = Mall + 176 * mid;
    double u = 0.0;
#pragma unroll
    for (int q = 0; q < 13; ++q) u += MB[row * 13 + q] * readlane_d(s2, q);
    double yk = 0.0;
#pragma unroll
    for (int q = 0; q < 13; ++q) yk += MB[q * 13 + row] * readlane_d(u, q);
    if (!cd_active(CD_B0 + 13 * mid + row, F, cmask)) yk = 0.0;
    if (lane < 13) { YB[13 * mid + lane] = yk; lds[E_XY + lane] = yk; }
    ymid = yk;
  }
  E_BARRIER();
  // backward sweeps, both halves at once:  down  y_k = M_k^T (u_k - T_A(k) y_{k-1})    k = m+1 .. F-1
  //                                        up    y_k = M_k^T (u_k - T'_A(k) y_{k+1})   k = m-1 .. 0
  double yprev = up ? lds[E_XY + row] : ymid;
  if (!up) {
    for (int k = mid + 1; k < F; ++k) {
      const double *TB = TAall + 176 * k, *MB = Mall + 176 * k;
      double s2 = U[13 * k + row];
#pragma unroll
      for (int q = 0; q < 13; ++q) s2 -= TB[row * 13 + q] * readlane_d(yprev, q);
      double yk = 0.0;
#pragma unroll
      for (int q = 0; q < 13; ++q) yk += MB[q * 13 + row] * readlane_d(s2, q);
      if (!cd_active(CD_B0 + 13 * k + row, F, cmask)) yk = 0.0;
      if (lane < 13) YB[13 * k + lane] = yk;
      yprev = yk;
    }
  } else {
    for (int k = mid - 1; k >= 0; --k) {
      const double *TB = TAall + 176 * k, *MB = Mall + 176 * k;
      double s2 = U[13 * k + row];
#pragma unroll
      for (int q = 0; q < 13; ++q) s2 -= TB[row * 13 + q] * readlane_d(yprev, q);
      double yk = 0.0;
#pragma unroll
      for (int q = 0; q < 13; ++q) yk += MB[q * 13 + row] * readlane_d(s2, q);
      if (!cd_active(CD_B0 + 13 * k + row, F, cmask)) yk = 0.0;
      if (lane < 13) YB[13 * k + lane] = yk;
      yprev = yk;
    }
  }
  lds_fence();
  // norms of this wave's frames
  for (int e = lane; e < 143; e += 64) {
    const int k = e / 13;
    if (k >= k_lo && k <= k_hi && k < F) {
      const double yb = YB[e];
      part_gnn += DB[e] * yb * yb;   // (y is zero on inactive dimensions)
      part_gy += GB[e] * yb;
    }
  }
  part_gnn = wave_sum(part_gnn); part_gy = wave_sum(part_gy);
  if (lane == 0) { red[QR_GNN + wave] = part_gnn; red[QR_GY + wave] = part_gy; red[QR_QX + wave] = 0.0; }
}

// B1 .. B4 after y_P: landmarks y_l = (g_l - w_l^T yP) / (E_l + mu dhat_l^2), a quarter each; partial sums to red[.. + wave].
__device__ __noinline__ void mw8_role_lm(int bw_, int wave_, const double *wl_, const double *lm_g_, const double *lm_einv_, const double *lm_dh2_, double *lm_y_, int L) {
  double *const lds = e_lds_base();
  const int bw = uni(bw_), wave = uni(wave_);
  const auto wl = uni_g(wl_), lm_g = uni_g(lm_g_), lm_einv = uni_g(lm_einv_), lm_dh2 = uni_g(lm_dh2_);
  const auto lm_y = uni_g(lm_y_);
  L = uni(L);
  const int lane = threadIdx.x & 63;
  double *g = lds + E_G, *dh2 = lds + E_DH2, *y = lds + E_Y, *red = lds + E_RED;
  double part_gnn = 0.0, part_gy = 0.0, part_qx = 0.0;
  const double *vP = lds + E_VP;
  for (int l = lane + 64 * bw; l < L; l += 256) {
    const double gl = lm_g[l], ei = lm_einv[l], d2 = lm_dh2[l], vl = gl / d2;
    double tl = 0.0, tq = 0.0;
    // (y and v_P through a pointer the compiler cannot see through: it would keep all 160 loop-invariant values in registers across the
    //  landmark loop, which runs once or twice, and leave 20 for the coupling entries in flight)
    const double *const yv = lds_in_vgpr(lds + E_Y), *const vPv = yv + (E_VP - E_Y);
    // (20 coupling entries in flight per lane; 40 — two round trips per landmark instead of four — measured slower: 83.5 against 82.6 us
    //  for one window, 97.8 against 94.9 us at 256 windows)
#pragma unroll 1
    for (int a0 = 0; a0 < 80; a0 += 20) {
      double wcol[20];
#pragma unroll
      for (int a = 0; a < 20; ++a) wcol[a] = wl[(size_t)(a0 + a) * L + l];
#pragma unroll
      for (int a = 0; a < 20; ++a)
        if (a0 + a < VILO_NPU) { tl += wcol[a] * yv[a0 + a]; tq += wcol[a] * vPv[a0 + a]; }   // y and v_P are zero on inactive dimensions
    }
    const double yl = (gl - tl) * ei;
    lm_y[l] = yl;
    if (l < 2 * ES_N) lds[E_WS + l] = yl;   // (the slices are done with: the candidate reads y_l here)
    part_gnn += d2 * yl * yl;
    part_gy += gl * yl;
    part_qx += vl * tq;   // cross term of q = v^T H v: 2 v_l w_l^T v_P
  }
  if (bw == 0) {
    for (int cd = lane; cd < 80; cd += 64) {
      part_gnn += dh2[cd] * y[cd] * y[cd];
      part_gy += g[cd] * y[cd];
    }
  }
  part_gnn = wave_sum(part_gnn); part_gy = wave_sum(part_gy); part_qx = wave_sum(part_qx);
  if (lane == 0) { red[QR_GNN + wave] = part_gnn; red[QR_GY + wave] = part_gy; red[QR_QX + wave] = part_qx; }
  E_BARRIER();   // (the chain waves' hand-over points)
  E_BARRIER();
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_solve_mw8(BatchDev b, SolveParams sp) {
  double *const lds = e_lds_base();
  const int win = blockIdx.x;
  SolverState &st = b.st[win];
  // (everything the kernel reads of the state and the window table before its first branch: one round trip, not one per question)
  const WinMeta wm = b.win[win];
  const int done0 = st.done, need_lin0 = st.need_lin, cur0 = st.cur, scale_ready0 = st.scale_ready;
  const double radius0 = st.radius, mu0 = st.mu;
  if (done0) return;
  // roles: 0: C1 (chain down + middle), 1: C2 (chain up), 2: TD, 3: TU, 4 .. 7: B1 .. B4. Hardware wave w of a workgroup runs on SIMD w % 4
  // (measured: waves w and w + 4 share a matrix pipe — one FP64 MFMA per 64 cycles per SIMD — and the older wave goes first): the chain waves
  // are paired with the T waves, the matrix waves with each other. (Pairing each matrix wave with a chain or T wave instead, which balances
  // the four pipes' instruction counts, changed nothing — 90.2 / 90.4 / 90.3 us for one window: the iterations are bound by the waves'
  // own dependency chains, not by the pipes.)
  const int hw_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifndef MW8_MAP
#define MW8_MAP 0
#endif
#if MW8_MAP == 0
  const int wave = (hw_wave & 2) ? (4 + (hw_wave & 1) + ((hw_wave & 4) >> 1)) : ((hw_wave & 1) + ((hw_wave & 4) >> 1));   // 0 1 4 5 | 2 3 6 7
#elif MW8_MAP == 1
  const int wave = hw_wave;
#else
  const int wave = hw_wave < 4 ? hw_wave : (hw_wave == 4 ? 6 : hw_wave == 6 ? 4 : hw_wave);
#endif
  const int lane = threadIdx.x & 63;
  const int tid = threadIdx.x;
  const int F = wm.n_frames, L = wm.L, kb = wm.pad, cmask = wm.const_mask;
  double *g = lds + E_G, *dh2 = lds + E_DH2, *y = lds + E_Y, *red = lds + E_RED;
  double *DB = lds + E_DB, *GB = lds + E_GB, *YB = lds + E_YB;
  struct DoglegIn { double radius, mu, gnorm2, gnnorm2, gdotgn, q, alpha, coef_a, coef_b, dogleg_step_norm, model_cost_change; int step_valid; } fresh;
  bool have_fresh = false;
  const long long q_start = pclk64();
  const double *x = b.x + (size_t)win * XSTRIDE;
  for (int e = tid; e < XSTRIDE; e += 512) lds[E_X + e] = x[e];   // (in flight behind everything below; read after many barriers)
  // the landmarks this thread scales and, at the end, moves: tid and tid + 512 keep lambda and g / dhat^2 in registers
  const double *lam = b.lam + wm.lm_off;
  double lam_r[2] = {0.0, 0.0}, vl_r[2] = {0.0, 0.0};
#pragma unroll
  for (int n = 0; n < 2; ++n) if (tid + 512 * n < wm.L) lam_r[n] = lam[tid + 512 * n];
  bool lmy_in_lds = false;
#define ESTAMP(w_, i_) PCLK(if (wave == (w_) && lane == 0) st.phase_clk[i_] = clock64() - q_start)

  if (need_lin0) {
    const double *bimg = b.Bimg + (size_t)win * BI_N;
    const double bscal0 = bimg[BI_SCAL + 0], bscal1 = bimg[BI_SCAL + 1], bscal2 = bimg[BI_SCAL + 2];
    const double *gin = b.cam_gin + (size_t)win * CD_N;
    const double *wl = b.lm_w + 80 * (size_t)wm.lm_off;
    double *lm_E = b.lm_E + wm.lm_off, *lm_g = b.lm_gbuf[cur0] + wm.lm_off, *lm_dh2 = b.lm_dh2 + wm.lm_off, *lm_scale = b.lm_scale + wm.lm_off,
           *lm_einv = b.lm_einv + wm.lm_off, *lm_y = b.lm_y + wm.lm_off;
    const bool first_scale = !scale_ready0;
    double mu = mu0;

    // ---- camera-side vectors come scaled from k_assemble; the landmarks are scaled here (landmark tid + 512 n) ----
    double part_gn = 0.0, part_gmax = 0.0, part_q = 0.0;
    if (wave == 4) {
      for (int cd = lane; cd < 80; cd += 64) { g[cd] = gin[cd]; dh2[cd] = bimg[BI_DH2 + cd]; lds[E_VP + cd] = cd_active(cd, F, cmask) ? bimg[BI_V + cd] : 0.0; }
    } else if (wave == 0) {
      for (int e = lane; e < 144; e += 64) { DB[e] = bimg[BI_DH2 + CD_B0 + min(e, 143)]; GB[e] = gin[CD_B0 + min(e, 143)]; }
    }
    for (int l = tid; l < L; l += 512) {
      const double E = lm_E[l], gl = lm_g[l];
      double sc;
      if (first_scale) { sc = sp.jacobi_scaling ? 1.0 / (1.0 + sqrt(E)) : 1.0; lm_scale[l] = sc; }
      else sc = lm_scale[l];
      const double d2 = fmin(fmax(sc * sc * E, sp.min_lm_diagonal), sp.max_lm_diagonal) / (sc * sc);
      lm_dh2[l] = d2;
      lm_einv[l] = 1.0 / (E + mu * d2);   // (of the first factorisation: a retry at a larger mu forms its own below)
      const double vl = gl / d2;
      if (l == tid) vl_r[0] = vl; else if (l == tid + 512) vl_r[1] = vl;
      part_q += E * vl * vl;   // (the cross term 2 vl w_l^T v_P comes from the back-substitution)
      part_gn += gl * vl;
      part_gmax = fmax(part_gmax, fabs(gl));
    }
    part_gn = wave_sum(part_gn); part_gmax = wave_max(part_gmax); part_q = wave_sum(part_q);
    if (lane == 0) { red[QR_GN + wave] = part_gn; red[QR_GMAX + wave] = part_gmax; red[QR_Q + wave] = part_q; }
    E_BARRIER_GLOBAL();
    double s_gn = 0.0, s_q = 0.0, s_gmax = 0.0;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) { s_gn += red[QR_GN + w8]; s_q += red[QR_Q + w8]; s_gmax = fmax(s_gmax, red[QR_GMAX + w8]); }
    const double gnorm2 = bscal1 + s_gn;
    const double gmax = fmax(bscal2, s_gmax);
    const double q_lm = s_q;
    if (!sp.fixed_iterations && gmax <= sp.gradient_tolerance) {
      if (tid == 0) { st.gmax = gmax; st.done = 1; st.termination = 1; st.step_valid = 0; }
      return;
    }

    ESTAMP(4, 0);   // prologue: scaling of the landmarks
    bool solved = false, retry = false;
    double qq = 0.0, gnnorm2 = 0.0, gy = 0.0;
    while (!solved) {
      if (retry) {
        for (int l = tid; l < L; l += 512) lm_einv[l] = 1.0 / (lm_E[l] + mu * lm_dh2[l]);
        E_BARRIER_GLOBAL();
      }
      retry = true;
      // ---- the roles of the factorisation (see the head of the file): every one meets 1 + NSTEP + 2 barriers ----
      if (wave < 2) { mw8_role_chain(wave, bimg, wl, lm_einv, lm_g, mu, F, L); ESTAMP(0, 8); }
      else if (wave < 4) mw8_role_T(wave - 2, bimg, wl, lm_einv, lm_g, F, L, kb);
      else {
        const double *Cimg = b.Cimg + (size_t)win * CIMG_N;
        const unsigned char *lms = b.lm_s + wm.lm_off;
        if (wave == 4) mw8_role_B<EB1_MASK, 0x03>(0, Cimg, lms, F, L);
        else if (wave == 5) mw8_role_B<EB2_MASK, 0x04>(1, Cimg, lms, F, L);
        else if (wave == 6) mw8_role_B<EB3_MASK, 0x08>(2, Cimg, lms, F, L);
        else mw8_role_B<EB4_MASK, 0x10>(3, Cimg, lms, F, L);
      }
      ESTAMP(4, 2);   // main loop (Schur complement + rank updates) and the tiles' hand-over
      {
        // the pose system: all eight waves (role c of mw8_chol80 = this wave's role in the main loop)
        const bool chain_failed = red[QR_FAIL] != 0.0 || red[QR_FAIL + 1] != 0.0;
        int f2 = chain_failed ? 1 : 0;
        if (!chain_failed) f2 = mw8_chol80(wave, mu, F, cmask);
        if (wave == 0 && lane == 0) red[QR_FAIL + 2] = (double)f2;
        ESTAMP(4, 4);   // Cholesky, forward and backward solve
        E_BARRIER();   // y_P (or the failure flags)
      }
      // (wave-uniform flags from LDS: chain down / up, pose system)
      if (red[QR_FAIL] != 0.0 || red[QR_FAIL + 1] != 0.0 || red[QR_FAIL + 2] != 0.0) {
        // DoglegStrategy::ComputeGaussNewtonStep: mu *= 10 and retry while mu < max_mu (1.0)
        mu *= 10.0;
        if (tid == 0) { st.mu = mu; st.pad[0]++; }   // (pad[0]: factorisation retries of this solve, read by the tests)
        if (!(mu < 1.0)) {
          if (tid == 0) { st.lin_fail = 1; st.step_valid = 0; st.gnorm2 = gnorm2; st.q = 0.0; st.gmax = gmax; st.scale_ready = 1; }
          return;
        }
        E_BARRIER();   // (every lane has read the flags before the next trip rewrites them)
        continue;
      }

      // ---- back-substitution: C1 / C2 the speed / leg-bias part (twisted sweeps), B1 .. B4 the landmarks; TD / TU keep the barriers company ----
      if (wave < 2) mw8_role_sweeps(wave, wave, bimg, F, L, kb, cmask);
      else if (wave < 4) {
        if (lane == 0) { red[QR_GNN + wave] = 0.0; red[QR_GY + wave] = 0.0; red[QR_QX + wave] = 0.0; }
        E_BARRIER();   // (the chain waves' hand-over points)
        E_BARRIER();
      } else mw8_role_lm(wave - 4, wave, wl, lm_g, lm_einv, lm_dh2, lm_y, L);
      ESTAMP(4, 5); ESTAMP(0, 9);   // back-substitutions
      // (the candidate takes lm_y from the LDS copy the landmark role leaves over the slices: no wait for the global stores — unless the
      //  window has more landmarks than the slices hold)
      if (L <= 2 * ES_N) E_BARRIER(); else E_BARRIER_GLOBAL();
      double s_gnn = 0.0, s_gy = 0.0, s_qx = 0.0;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) { s_gnn += red[QR_GNN + w8]; s_gy += red[QR_GY + w8]; s_qx += red[QR_QX + w8]; }
      gnnorm2 = s_gnn;
      gy = s_gy;
      qq = bscal0 + (q_lm + 2.0 * s_qx);
      if (!(isfinite(gnnorm2) && isfinite(gy))) {   // IsArrayValid(gauss_newton_step_) failed
        mu *= 10.0;
        if (tid == 0) { st.mu = mu; st.pad[0]++; }
        if (!(mu < 1.0)) {
          if (tid == 0) { st.lin_fail = 1; st.step_valid = 0; st.scale_ready = 1; }
          return;
        }
        E_BARRIER();
        continue;
      }
      solved = true;
    }
    lmy_in_lds = L <= 2 * ES_N;
    // keep the linearisation's vectors for the steps that reuse it after a rejected candidate
    double *cam_g = b.cam_g + (size_t)win * CD_N, *cam_dh2 = b.cam_dh2 + (size_t)win * CD_N, *cam_y = b.cam_y + (size_t)win * CD_N;
    if (wave == 4) {
      for (int cd = lane; cd < 80; cd += 64) { cam_g[cd] = g[cd]; cam_dh2[cd] = dh2[cd]; cam_y[cd] = y[cd]; }
    } else if (wave == 0) {
      for (int e = lane; e < 144; e += 64) { cam_g[CD_B0 + e] = GB[e]; cam_dh2[CD_B0 + e] = DB[e]; cam_y[CD_B0 + e] = (e < 13 * F) ? YB[e] : 0.0; }
    }
    if (tid == 0) {
      st.gnorm2 = gnorm2; st.gnnorm2 = gnnorm2; st.gdotgn = -gy; st.q = qq; st.gmax = gmax;
      st.alpha = gnorm2 / qq;
      st.scale_ready = 1;
      st.lin_fail = 0;
      // (the dogleg below works on a register copy of these: no round trip through what was just stored)
      fresh.gnorm2 = gnorm2; fresh.gnnorm2 = gnnorm2; fresh.gdotgn = -gy; fresh.q = qq; fresh.alpha = gnorm2 / qq; fresh.mu = mu;
      have_fresh = true;
    }
  } else {
    const double *cam_g = b.cam_g + (size_t)win * CD_N, *cam_dh2 = b.cam_dh2 + (size_t)win * CD_N, *cam_y = b.cam_y + (size_t)win * CD_N;
    if (wave == 4) {
      for (int cd = lane; cd < 80; cd += 64) { g[cd] = cam_g[cd]; dh2[cd] = cam_dh2[cd]; y[cd] = cam_y[cd]; }
    } else if (wave == 0) {
      for (int e = lane; e < 144; e += 64) { GB[e] = cam_g[CD_B0 + min(e, 143)]; DB[e] = cam_dh2[CD_B0 + min(e, 143)]; YB[e] = cam_y[CD_B0 + min(e, 143)]; }
    }
  }

  ESTAMP(4, 6);   // norms, vectors kept
  // ---- dogleg step for the current radius, candidate camera state ----
  if (tid == 0) {
    double ca = 0.0, cb = 0.0;
    int go = 0;
    const double radius = radius0;   // (read when the kernel started)
    if (radius <= sp.min_radius) { st.done = 1; st.termination = 1; st.step_valid = 0; }
    else if (have_fresh) {
      fresh.radius = radius;
      dogleg_scalars(fresh);
      st.coef_a = fresh.coef_a; st.coef_b = fresh.coef_b; st.dogleg_step_norm = fresh.dogleg_step_norm; st.model_cost_change = fresh.model_cost_change;
      st.step_valid = fresh.step_valid;
      ca = fresh.coef_a; cb = fresh.coef_b; go = fresh.step_valid;
    }
    else { dogleg_scalars(st); ca = st.coef_a; cb = st.coef_b; go = st.step_valid; }
    red[QR_CA] = ca; red[QR_CB] = cb; red[QR_GO] = (double)go;
  }
  E_BARRIER();
  const double ca = red[QR_CA], cb = red[QR_CB];
  const int go = (red[QR_GO] != 0.0) ? 1 : 0;
  double *xc = b.xc + (size_t)win * XSTRIDE;
  const double *xl = lds + E_X;   // (the state as fetched at the start)
  {
    const double *lmg = b.lm_gbuf[cur0] + wm.lm_off, *lmd = b.lm_dh2 + wm.lm_off, *lmy = b.lm_y + wm.lm_off;
    double *lamc = b.lamc + wm.lm_off;
    for (int l = tid, n = 0; l < L; l += 512, ++n) {
      const double la = n < 2 ? lam_r[n < 1 ? 0 : 1] : lam[l];
      if (!go) { lamc[l] = la; continue; }
      const double vl = (need_lin0 && n < 2) ? vl_r[n < 1 ? 0 : 1] : lmg[l] / lmd[l];
      const double yl = lmy_in_lds ? lds[E_WS + l] : lmy[l];
      lamc[l] = la - ca * vl - cb * yl;
    }
  }
  if (!go) {
    for (int e = tid; e < XSTRIDE; e += 512) xc[e] = xl[e];
    return;
  }
  double *del = lds + E_DEL;
  if (wave == 4) {
    for (int cd = lane; cd < 80; cd += 64) del[cd] = -ca * g[cd] / dh2[cd] - cb * y[cd];
  } else if (wave == 0) {
    for (int e = lane; e < 143; e += 64) del[CD_B0 + e] = -ca * GB[e] / DB[e] - cb * YB[e];
  }
  E_BARRIER();
  if (wave == 4) {
    if (lane < 11) pose_plus(xl + XO_POSE + 7 * lane, del + 6 * lane, xc + XO_POSE + 7 * lane);
    else if (lane < 13) pose_plus(xl + XO_EX + 7 * (lane - 11), del + CD_EX0 + 6 * (lane - 11), xc + XO_EX + 7 * (lane - 11));
    else if (lane == 13) xc[XO_TD] = xl[XO_TD] + del[CD_TD];
  } else if (wave == 0) {
    for (int e = lane; e < 143; e += 64) {
      const int k = e / 13, c = e - 13 * k;
      if (c < 9) xc[XO_SB + 9 * k + c] = xl[XO_SB + 9 * k + c] + del[CD_B0 + e];
      else xc[XO_LB + 4 * k + (c - 9)] = xl[XO_LB + 4 * k + (c - 9)] + del[CD_B0 + e];
    }
  }
  ESTAMP(4, 7);   // dogleg + candidate
  PCLK(if (tid == 0) { for (int r_ = 0; r_ < 8; ++r_) { st.phase_clk[16 + r_] = (long long)red[QR_WAIT + r_]; st.phase_clk[24 + r_] = (long long)red[64 + r_]; } st.phase_clk[10] = (long long)red[72]; st.phase_clk[11] = (long long)red[73]; for (int r_ = 0; r_ < 11; ++r_) st.phase_clk[47 + r_] = (long long)red[76 + r_]; });
}

int vilo_launch_mw8_solver(vilo_ctx *ctx, BatchDev &b, const SolveParams &sp, hipStream_t s) {
  const size_t lds_bytes = (size_t)E_TOTAL * sizeof(double);
  if (!ctx->mw8_attr_set) {   // (per context = per device, like the other dynamic-LDS opt-ins)
    VILO_HIP(hipFuncSetAttribute((const void *)k_solve_mw8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    ctx->mw8_attr_set = true;
  }
  hipLaunchKernelGGL(k_solve_mw8, dim3(b.W), dim3(512), lds_bytes, s, b, sp);
  return VILO_OK;
}
