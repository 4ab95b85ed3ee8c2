// Lane-parallel construction of dF = F - I and V of one midpoint step of IMULegIntegrationBase (imu_leg_integration_base.cpp:376-465).
//
// The blocks of F and V are sums of a few 3 x 3 products of the step's rotations with per-sample matrices, scaled by powers of dt. One lane
// doing that with 3 x 3 temporaries in registers issues ~900 FP64 instructions per step (the whole wave waits on it); here every product
// entry and every output entry is one lane's work:
//   pool     3 x 3 matrices (row-major, 9 doubles) in LDS: the step's R_0, R_1, kappa_7, [a_0]x, [a_1]x, [w]x, the constants R_br and I, and
//            per (leg, endpoint) [v]x, [p_br + R_br f]x, h_0, J (from the interval's precomputed sample records)
//   level 1  28 products  R_e * X                       (4 rounds of 7 products x 9 entries = 63 lanes)
//   level 2  13 products  (R_1 [a_1]x) kappa_7, (R_1 [v_1]x) kappa_7, (R_e R_br) J_e    (2 rounds)
//   blocks   57 blocks of F and V, each  sum_k coef[c_k] * pool[s_k]  entry by entry (9 rounds)
// The tables are compile-time constants; a lane keeps its 6 + 9 descriptors in registers for the whole interval.
// Host-compilable: tests/host_check emulates the lanes and compares with the blocks written out as in the reference.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PB_HD __host__ __device__ __forceinline__
#else
#define PB_HD inline
#endif

namespace vilo {
namespace pb {

// ---- LDS map of the step (doubles) ----
constexpr int FCLD = 17;    // dF has non-zero columns only at K = {3 .. 8, 21 .. 30}: stored compact, column kidx(c) (order: below), odd leading dimension
constexpr int VLD = 49;     // V: 32 x 48 (46 noise dimensions), odd leading dimension
constexpr int O_FC = 0;                      // [32 x 17]
constexpr int O_VM = O_FC + 32 * FCLD;       // [32 x 49]
constexpr int O_POOL = O_VM + 32 * VLD;      // [N_SLOTS x 9]
constexpr int N_SLOTS = 69;                    // (the last one: where idle lanes write)
constexpr int O_GIN = O_POOL + 9 * N_SLOTS;  // [8 x 3] g_0 of (leg, endpoint)
constexpr int O_GV = O_GIN + 24;             // [8 x 3] -(R_e g_0)
constexpr int O_VV = O_GV + 24;              // [8 x 3] v of (leg, endpoint)
constexpr int O_LOV = O_VV + 24;             // [8 x 3] q_e v of (leg, endpoint): the halves of the leg-odometry velocity
constexpr int O_FF = O_LOV + 24;             // [36] contact-type-2 force filter: min, max, variance (4 each), window (4 x 5), index (4)
constexpr int O_COEF = O_FF + 36;            // [16]
constexpr int O_ND = O_COEF + 16;            // [48] noise diagonal
constexpr int O_DUMMY = O_ND + 48;            // [9] where idle lanes write
constexpr int PB_TOTAL = O_DUMMY + 9;

// Compact order of K: first the eight columns whose entries all lie in rows 0 .. 15 — d/dv (6 .. 8: the dt I block of the position rows),
// d/dba (21 .. 23: position and velocity rows), d/drho of legs 0 and 1 (27, 28: their epsilon rows 9 .. 14) —, then theta (3 .. 5), bg
// (24 .. 26) and the calf lengths of legs 2 and 3 (29, 30), which reach into rows 16 .. 20 as well: the first two k-steps of F X and Q F^T
// have nothing in one half of dF's rows, and kernels_preint.hip issues half of their products (DF_KSTEPS_ROWS_LO_ONLY).
PB_HD constexpr int kidx(int c) {     // column of F -> compact column (c in K)
  return c < 6 ? c + 5 : c < 9 ? c - 6 : c < 24 ? c - 18 : c < 27 ? c - 13 : c < 29 ? c - 21 : c - 15;
}
PB_HD constexpr int fk_col(int k) {   // and back: 6 7 8 21 | 22 23 27 28 | 3 4 5 24 | 25 26 29 30
  return k < 3 ? 6 + k : k < 6 ? 18 + k : k < 8 ? 21 + k : k < 11 ? k - 5 : k < 14 ? 13 + k : 15 + k;
}
constexpr unsigned DF_KSTEPS_ROWS_LO_ONLY = 0x3u;
static_assert(fk_col(kidx(3)) == 3 && fk_col(kidx(8)) == 8 && fk_col(kidx(21)) == 21 && fk_col(kidx(26)) == 26 && fk_col(kidx(27)) == 27 && fk_col(kidx(30)) == 30, "K");
static_assert(kidx(4) == kidx(3) + 1 && kidx(5) == kidx(3) + 2 && kidx(7) == kidx(6) + 1 && kidx(8) == kidx(6) + 2 && kidx(22) == kidx(21) + 1 && kidx(23) == kidx(21) + 2 &&
              kidx(25) == kidx(24) + 1 && kidx(26) == kidx(24) + 2, "the 3 x 3 blocks' columns stay adjacent");

// ---- pool slots ----
constexpr int S_R0 = 0, S_R1 = 1, S_K7 = 2, S_RA0 = 3, S_RA1 = 4, S_RWX = 5, S_RBR = 6, S_I = 7;
constexpr int S_SKV = 8, S_SKP = 16, S_H0 = 24, S_J = 32;        // + 2 * leg + endpoint: the sample records land here (REC_MATS consecutive groups of 8)
constexpr int S_M1 = 40, S_M2 = 41, S_RB0 = 42, S_RB1 = 43;      // R_0 [a_0]x, R_1 [a_1]x, R_0 R_br, R_1 R_br
constexpr int S_S = 44, S_P = 52, S_H = 60;                      // + 2 * leg + endpoint: R_e [v]x, R_e [p]x, R_e h_0
// second level and pool blocks, in the slots of first-level inputs that are dead by then
constexpr int S_M3 = 8;       // (R_1 [a_1]x) kappa_7
constexpr int S_S1K = 9;      // + leg: (R_1 [v_1]x) kappa_7
constexpr int S_G = 16;       // + 2 * leg + endpoint: (R_e R_br) J_e
constexpr int S_DUMMY = 68;

// ---- sample record (per sample and leg, written once per interval): what the pool needs, in pool order ----
constexpr int REC_MATS = 4;                  // [v]x, [p_br + R_br f]x, h_0, J
constexpr int REC_N = 9 * REC_MATS + 6;      // + g_0 (3) + v (3) = 42 doubles

// ---- coefficients (functions of dt, written out as the reference writes them) ----
constexpr int C_ONE = 0, C_DT = 1, C_MDT = 2, C_HDT = 3, C_MHDT = 4, C_QDT2 = 5, C_MQDT2 = 6, C_HDT2 = 7, C_QDT3 = 8, C_MEDT3 = 9, C_ZERO = 15;
PB_HD void coefficients(double dt, double *c) {
  c[C_ONE] = 1.0; c[C_DT] = dt; c[C_MDT] = -dt; c[C_HDT] = 0.5 * dt; c[C_MHDT] = -0.5 * dt;
  c[C_QDT2] = 0.25 * dt * dt; c[C_MQDT2] = -0.25 * dt * dt; c[C_HDT2] = 0.5 * dt * dt;
  c[C_QDT3] = 0.25 * dt * dt * dt; c[C_MEDT3] = -0.25 * dt * dt * 0.5 * dt;
  c[10] = c[11] = c[12] = c[13] = c[14] = 0.0; c[C_ZERO] = 0.0;
}

// ---- descriptors ----
// product: A | B << 8 | D << 16;   pool[D] = pool[A] * pool[B]
// block:   offset of its (0, 0) entry in the step's LDS array | ld << 16 | (coef | slot << 4) << 22 / 33 / 44
// A round has 8 descriptors: 7 for the lanes' groups of 9, the eighth (and every unused one) a no-op into the dummy slot — no branches.
constexpr int N_PROD1 = 28, N_PROD2 = 13, N_PROD_ROUNDS = 6, N_BLK_ROUNDS = 9;
struct Tables {
  unsigned prod[8 * N_PROD_ROUNDS];
  unsigned long long blk[8 * N_BLK_ROUNDS];
};
constexpr unsigned mk_prod(int A, int B, int D) { return (unsigned)A | ((unsigned)B << 8) | ((unsigned)D << 16); }
constexpr unsigned long long mk_term(int c, int s) { return (unsigned long long)c | ((unsigned long long)s << 4); }
constexpr unsigned long long mk_blk(int off, int ld, int c0, int s0, int c1 = C_ZERO, int s1 = S_I, int c2 = C_ZERO, int s2 = S_I) {
  return (unsigned long long)off | ((unsigned long long)ld << 16) | (mk_term(c0, s0) << 22) | (mk_term(c1, s1) << 33) | (mk_term(c2, s2) << 44);
}
constexpr int fc_off(int r, int c) { return O_FC + r * FCLD + kidx(c); }
constexpr int vm_off(int r, int c) { return O_VM + r * VLD + c; }
// k-steps (4 columns each) of V whose non-zero rows all lie in 0 .. 15 / in 16 .. 31 (from the block table below; checked on the host by
// tests/test_host_math.py against the table itself)
constexpr unsigned V_KSTEPS_ROWS_LO_ONLY = 1u << 8;
constexpr unsigned V_KSTEPS_ROWS_HI_ONLY = (1u << 3) | (1u << 10) | (1u << 11);

constexpr Tables make_tables() {
  Tables t{};
  for (int i = 0; i < 8 * N_PROD_ROUNDS; ++i) t.prod[i] = mk_prod(S_I, S_I, S_DUMMY);
  for (int i = 0; i < 8 * N_BLK_ROUNDS; ++i) t.blk[i] = mk_blk(O_DUMMY, 3, C_ZERO, S_I);
  // (filled in order below, then spread to 8 per round at the end)
  int n = 0;
  // level 1 (rounds 0 .. 3)
  t.prod[n++] = mk_prod(S_R0, S_RA0, S_M1);
  t.prod[n++] = mk_prod(S_R1, S_RA1, S_M2);
  t.prod[n++] = mk_prod(S_R0, S_RBR, S_RB0);
  t.prod[n++] = mk_prod(S_R1, S_RBR, S_RB1);
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 2; ++e) {
      const int R = e ? S_R1 : S_R0, q = 2 * j + e;
      t.prod[n++] = mk_prod(R, S_SKV + q, S_S + q);
      t.prod[n++] = mk_prod(R, S_SKP + q, S_P + q);
      t.prod[n++] = mk_prod(R, S_H0 + q, S_H + q);
    }
  // level 2 (rounds 4, 5)
  t.prod[n++] = mk_prod(S_M2, S_K7, S_M3);
  for (int j = 0; j < 4; ++j) t.prod[n++] = mk_prod(S_S + 2 * j + 1, S_K7, S_S1K + j);
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 2; ++e) t.prod[n++] = mk_prod(e ? S_RB1 : S_RB0, S_J + 2 * j + e, S_G + 2 * j + e);
  // blocks. F - I, IMU rows (:382-399); kappa_1 = (R_0 [a_0]x) (-dt / 2) + (R_1 [a_1]x kappa_7) (-dt / 2) (:380) and R_0 + R_1 are
  // sums of two pool matrices, so the blocks that scale them take both terms with the product of the scalings
  int m = 0;
  t.blk[m++] = mk_blk(fc_off(0, 3), FCLD, C_MQDT2, S_M1, C_MQDT2, S_M3);       // kappa_1 dt / 2
  t.blk[m++] = mk_blk(fc_off(0, 6), FCLD, C_DT, S_I);
  t.blk[m++] = mk_blk(fc_off(0, 21), FCLD, C_MQDT2, S_R0, C_MQDT2, S_R1);
  t.blk[m++] = mk_blk(fc_off(0, 24), FCLD, C_QDT3, S_M2);
  t.blk[m++] = mk_blk(fc_off(3, 3), FCLD, C_MDT, S_RWX);
  t.blk[m++] = mk_blk(fc_off(3, 24), FCLD, C_MDT, S_I);
  t.blk[m++] = mk_blk(fc_off(6, 3), FCLD, C_MHDT, S_M1, C_MHDT, S_M3);         // kappa_1
  t.blk[m++] = mk_blk(fc_off(6, 21), FCLD, C_MHDT, S_R0, C_MHDT, S_R1);
  t.blk[m++] = mk_blk(fc_off(6, 24), FCLD, C_HDT2, S_M2);
  // V, IMU rows (:420-438)
  t.blk[m++] = mk_blk(vm_off(0, 0), VLD, C_QDT2, S_R0);
  t.blk[m++] = mk_blk(vm_off(0, 3), VLD, C_MEDT3, S_M2);
  t.blk[m++] = mk_blk(vm_off(0, 6), VLD, C_QDT2, S_R1);
  t.blk[m++] = mk_blk(vm_off(0, 9), VLD, C_MEDT3, S_M2);
  t.blk[m++] = mk_blk(vm_off(3, 3), VLD, C_HDT, S_I);
  t.blk[m++] = mk_blk(vm_off(3, 9), VLD, C_HDT, S_I);
  t.blk[m++] = mk_blk(vm_off(6, 0), VLD, C_HDT, S_R0);
  t.blk[m++] = mk_blk(vm_off(6, 3), VLD, C_MQDT2, S_M2);
  t.blk[m++] = mk_blk(vm_off(6, 6), VLD, C_HDT, S_R1);
  t.blk[m++] = mk_blk(vm_off(6, 9), VLD, C_MQDT2, S_M2);
  t.blk[m++] = mk_blk(vm_off(21, 12), VLD, C_MDT, S_I);
  t.blk[m++] = mk_blk(vm_off(24, 15), VLD, C_MDT, S_I);
  // the rows of leg j (:401-417, :440-461)
  for (int j = 0; j < 4; ++j) {
    const int e = 9 + 3 * j;
    t.blk[m++] = mk_blk(fc_off(e, 3), FCLD, C_MHDT, S_S + 2 * j, C_MHDT, S_S1K + j);
    t.blk[m++] = mk_blk(fc_off(e, 24), FCLD, C_HDT2, S_S + 2 * j + 1, C_MHDT, S_P + 2 * j, C_MHDT, S_P + 2 * j + 1);
    t.blk[m++] = mk_blk(vm_off(e, 3), VLD, C_MQDT2, S_S + 2 * j + 1, C_HDT, S_P + 2 * j);
    t.blk[m++] = mk_blk(vm_off(e, 9), VLD, C_MQDT2, S_S + 2 * j + 1, C_HDT, S_P + 2 * j + 1);
    t.blk[m++] = mk_blk(vm_off(e, 18), VLD, C_MHDT, S_H + 2 * j);
    t.blk[m++] = mk_blk(vm_off(e, 21), VLD, C_MHDT, S_H + 2 * j + 1);
    t.blk[m++] = mk_blk(vm_off(e, 24), VLD, C_MHDT, S_G + 2 * j);
    t.blk[m++] = mk_blk(vm_off(e, 27), VLD, C_MHDT, S_G + 2 * j + 1);
    t.blk[m++] = mk_blk(vm_off(e, 30 + 3 * j), VLD, C_MDT, S_I);
  }
  // spread: entry 7 r + g -> 8 r + g (from the back; slot 8 r + 7 stays the no-op)
  for (int r = N_PROD_ROUNDS - 1; r >= 0; --r)
    for (int g = 6; g >= 0; --g) {
      const unsigned v = (7 * r + g < N_PROD1 + N_PROD2 + 1) ? t.prod[7 * r + g] : mk_prod(S_I, S_I, S_DUMMY);
      t.prod[8 * r + g] = v;
    }
  for (int r = 0; r < N_PROD_ROUNDS; ++r) t.prod[8 * r + 7] = mk_prod(S_I, S_I, S_DUMMY);
  for (int r = N_BLK_ROUNDS - 1; r >= 0; --r)
    for (int g = 6; g >= 0; --g) {
      const unsigned long long v = (7 * r + g < m) ? t.blk[7 * r + g] : mk_blk(O_DUMMY, 3, C_ZERO, S_I);
      t.blk[8 * r + g] = v;
    }
  for (int r = 0; r < N_BLK_ROUNDS; ++r) t.blk[8 * r + 7] = mk_blk(O_DUMMY, 3, C_ZERO, S_I);
  return t;
}
static_assert(N_PROD1 + N_PROD2 <= 7 * N_PROD_ROUNDS - 1, "product table");
static_assert(21 + 36 <= 7 * N_BLK_ROUNDS, "block table");

// ---- one lane's share of a round ----
// entry ent (0 .. 8) of pool[D] = pool[A] * pool[B]: a_i0 b_0j + a_i1 b_1j + a_i2 b_2j, summed in that order
PB_HD void product_entry(unsigned d, int ent, double *L) {
  const int A = d & 0xff, B = (d >> 8) & 0xff, D = (d >> 16) & 0xff;
  const int i = ent / 3, j = ent - 3 * i;
  const double *a = L + O_POOL + 9 * A + 3 * i, *b = L + O_POOL + 9 * B + j;
  L[O_POOL + 9 * D + ent] = a[0] * b[0] + a[1] * b[3] + a[2] * b[6];
}
// entry ent of a block: sum_k coef[c_k] * pool[s_k][ent]
PB_HD void block_entry(unsigned long long d, int ent, double *L) {
  const int off = (int)(d & 0xffff), ld = (int)((d >> 16) & 0x3f);
  const int c0 = (int)((d >> 22) & 15), s0 = (int)((d >> 26) & 127), c1 = (int)((d >> 33) & 15), s1 = (int)((d >> 37) & 127),
            c2 = (int)((d >> 44) & 15), s2 = (int)((d >> 48) & 127);
  const int i = ent / 3, j = ent - 3 * i;
  const double *coef = L + O_COEF, *pool = L + O_POOL;
  double v = coef[c0] * pool[9 * s0 + ent];
  v += coef[c1] * pool[9 * s1 + ent];
  v += coef[c2] * pool[9 * s2 + ent];
  L[off + i * ld + j] = v;
}
// lanes 0 .. 23: -(R_e g_0) of (leg, endpoint) q = lane / 3, row lane % 3   (:262, :283: g = -(R g_0))
PB_HD void gvec_entry(int lane, double *L) {
  if (lane >= 24) return;
  const int q = lane / 3, i = lane - 3 * q;
  const double *R = L + O_POOL + 9 * ((q & 1) ? S_R1 : S_R0) + 3 * i, *g = L + O_GIN + 3 * q;
  L[O_GV + lane] = -(R[0] * g[0] + R[1] * g[1] + R[2] * g[2]);
}
// lanes 0 .. 11: F(9 + 3 j + a, 27 + j) = (g_j0 + g_j1) dt / 2 (:408);  lanes 12 .. 15: V(27 + j, 42 + j) = -dt (:463)
PB_HD void tail_entry(int lane, double dt, double *L) {
  if (lane < 12) {
    const int j = lane / 3, a = lane - 3 * j;
    L[fc_off(9 + 3 * j + a, 27 + j)] = (L[O_GV + 6 * j + a] + L[O_GV + 6 * j + 3 + a]) * (0.5 * dt);
  } else if (lane < 16) {
    const int j = lane - 12;
    L[vm_off(27 + j, 42 + j)] = -dt;
  }
}
// (the callers keep the quaternion rotation of the leg velocities, q_e v, in vilo_math's qrot: lanes 0 .. 7 store it at O_LOV + 3 lane)
// where element r (0 .. REC_N - 1) of the sample record of (leg j, endpoint e) goes in the step's LDS array
PB_HD int record_dest(int j, int e, int r) {
  const int q = 2 * j + e;
  if (r < 9 * REC_MATS) { const int mth = r / 9; return O_POOL + 9 * (S_SKV + 8 * mth + q) + (r - 9 * mth); }
  if (r < 9 * REC_MATS + 3) return O_GIN + 3 * q + (r - 9 * REC_MATS);
  return O_VV + 3 * q + (r - 9 * REC_MATS - 3);
}

}  // namespace pb
}  // namespace vilo
