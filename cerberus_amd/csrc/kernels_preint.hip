// Contact (IMU + leg) preintegration on gfx950: IMULegIntegrationBase's constructor + push_back()/propagate()/
// midPointIntegration() (imu_leg_integration_base.cpp:7-470) and the classic IntegrationBase
// (integration_base.h:18-170), batched over intervals, samples sequential (each step depends on the previous one).
// IMU-leg (31 states): a producer / consumer pair of waves per interval — dF = F - I and V built lane-parallel in LDS by one wave,
// jacobian <- F jacobian, covariance <- F covariance F^T + V N V^T on the FP64 matrix cores by the other (preint_imu_leg_body).
// IMU only (15 states): one wave per interval, the updates parallel over columns (lane = column), everything resident in LDS.
//
// The batch form, the streaming form and the re-propagation of a resident batch are three instantiations of one body and must give
// bitwise the same record (tests: streaming == batch). With the default -ffp-contract=fast the optimiser fuses a multiply with an
// add wherever inlining happens to bring them together, which differs between the instantiations; contract(on) fuses inside one
// source expression only — decided by the front end, the same in every instantiation.
#pragma clang fp contract(on)
#include "solver_types.hpp"
#include "preint_blocks.hpp"

using namespace vilo;

// A device-resident IMULegIntegrationBase (imu_leg_integration_base.h:73-128): public state, the previous sample
// (acc_0, gyr_0, phi_0, dphi_0, c_0) and the contact-type-2 force filter
struct PreintStream {
  vilo_preint rec;
  vilo_sample last;
  double ff_min[4], ff_max[4], ff_win[20], ff_var[4];
  int ff_idx[4];
  int n_pushed, pad;
};

// the same for an IntegrationBase (integration_base.h:201-220), the USE_LEG = 0 configurations
struct PreintImuStream {
  vilo_preint_imu rec;
  vilo_sample last;
  int n_pushed, pad;
};

__constant__ const pb::Tables c_pb_tab = pb::make_tables();

namespace {

// jacobian <- F jacobian ; covariance <- F covariance F^T + V diag(nd) V^T   (:467-468 / integration_base.h:136-137)
template <int NS, int NN>
__device__ void jac_cov_update(const double *Fm, const double *Vm, const double *nd, double *Jm, double *Pm, double *Qm) {
  const int c = threadIdx.x;
  constexpr int LD = NS + 1;
  double col[NS];
  if (c < NS) {
    for (int r = 0; r < NS; ++r) {
      double s = 0.0;
      for (int k = 0; k < NS; ++k) s += Fm[r * NS + k] * Jm[k * LD + c];
      col[r] = s;
    }
  }
  __syncthreads();
  if (c < NS)
    for (int r = 0; r < NS; ++r) Jm[r * LD + c] = col[r];
  // Q = F P
  if (c < NS) {
    for (int r = 0; r < NS; ++r) {
      double s = 0.0;
      for (int k = 0; k < NS; ++k) s += Fm[r * NS + k] * Pm[k * LD + c];
      Qm[r * LD + c] = s;
    }
  }
  __syncthreads();
  // P' = Q F^T + V N V^T, column c
  if (c < NS) {
    for (int r = 0; r < NS; ++r) {
      double s = 0.0;
      for (int k = 0; k < NS; ++k) s += Qm[r * LD + k] * Fm[c * NS + k];
      double t = 0.0;
      for (int k = 0; k < NN; ++k) t += Vm[r * NN + k] * nd[k] * Vm[c * NN + k];
      Pm[r * LD + c] = s + t;
    }
  }
  __syncthreads();
}

// The same update for the 31-dim IMU-leg state on the FP64 matrix cores: all four products are 32 x 32 (x 32 or x 48) GEMMs of
// zero-padded LDS matrices, 2 x 2 output tiles of v_mfma_f64_16x16x4 (lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16] of a
// 4-deep k-step and owns C[(l / 16) + 4 r][l % 16], r = 0..3).
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
constexpr int FLD = 33;        // leading dimension of the 32 x 32 matrices (odd: rows land in different LDS banks)
constexpr int VLD = pb::VLD;   // of the 32 x 48 noise Jacobian
constexpr int FCLD = pb::FCLD; // of dF = F - I, stored with its 16 non-zero columns only (preint_blocks.hpp)

// F = I + dF, and dF has non-zero columns only at K = {3 .. 8, 21 .. 30} (d/d theta, d/d v, d/d ba, d/d bg, d/d rho): 16 of 31
// (compact order: preint_blocks.hpp — the two k-steps whose columns live in rows 0 .. 15 only come first and cost half the products).
// With dFc = (F - I)[:, K] (32 x 16, leading dimension FCLD) the products keep their identity part in the accumulators and contract
// over the 16 columns of K only:
//   F X = X + dF[:, K] X[K, :],   Q F^T = Q + Q[:, K] dF[:, K]^T        (4 k-steps instead of 8 each),
// and V N V^T has four of its 12 k-steps (4 noise columns each) in one half of the rows only (preint_blocks.hpp: columns 12 .. 15 are the
// bias random walks' rows 21 .. 26; 32 .. 35 the foot-position noise of legs 0 / 1, rows 9 .. 14; 40 .. 47 leg 3's, rows 18 .. 20, the calf
// lengths' rows 27 .. 30 and padding): three of such a step's four products are exact zeros and are not issued. 78 matrix instructions
// per sample: jacobian <- F jacobian ; covariance <- F covariance F^T + V diag(nd) V^T   (imu_leg_integration_base.cpp:467-468).
//
// The consumer wave of the pair below takes the step's dF and V operands out of LDS in one go, so that the producer can build the next
// step's blocks while these products run.
constexpr int BLK_ROUNDS_PRODUCER = 5;   // of pb::N_BLK_ROUNDS rounds of block entries: the producer's; the others and the tail are the consumer's
constexpr int PAIR_THREADS = 128;   // the IMU-leg kernels' workgroup: producer + consumer wave (preint_imu_leg_body)
struct FvOperands {
  double f0[4], f1[4];     // dF[lr][4 kk + lk], dF[16 + lr][4 kk + lk] (compact columns): the A operand of dF X and the B operand of Q dF^T
  double v0[12], v1[12];   // V[lr][4 kk + lk], V[16 + lr][4 kk + lk]: both operands of V N V^T
  double sc[12];           // nd[4 kk + lk]
};
__device__ __forceinline__ void load_fv_operands(const double *dFm, const double *Vm, const double *nd, FvOperands &o) {
  const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const bool hi = !((pb::DF_KSTEPS_ROWS_LO_ONLY >> kk) & 1u);
    o.f0[kk] = dFm[lr * FCLD + 4 * kk + lk];
    o.f1[kk] = hi ? dFm[(16 + lr) * FCLD + 4 * kk + lk] : 0.0;
  }
#pragma unroll
  for (int kk = 0; kk < 12; ++kk) {
    const bool lo = !((pb::V_KSTEPS_ROWS_HI_ONLY >> kk) & 1u), hi = !((pb::V_KSTEPS_ROWS_LO_ONLY >> kk) & 1u);
    o.v0[kk] = lo ? Vm[lr * VLD + 4 * kk + lk] : 0.0;
    o.v1[kk] = hi ? Vm[(16 + lr) * VLD + 4 * kk + lk] : 0.0;
    o.sc[kk] = nd[4 * kk + lk];
  }
}
// a wave-uniform double, moved to scalar registers
__device__ __forceinline__ double uni_d(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
// The "blocks ready" rendezvous of a producer / consumer pair. The two roles reach it at different program sites (each inside its own loop
// over the interval's steps), so it is spelled as what it is, the hardware's arrival-counting s_barrier between workgroup-scope fences,
// and not as __syncthreads(), whose contract is that all threads reach the same call. Invariant: both roles execute it exactly
// s_end - si_first times (once a step) before they leave their loops; the kernel is built for gfx950 only.
__device__ __forceinline__ void pair_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void wave_fence() { asm volatile("" ::: "memory"); }   // (one wave's LDS operations complete in program order: only the compiler must keep it)
// The consumer's LDS keeps of the 32 x 32 jacobian / covariance only what the next product reads (the matrices themselves live in its
// accumulators from one step to the next): the rows K = {3 .. 8, 21 .. 30} as the B operand of dF X[K, :], the columns K of Q as the A
// operand of Q[:, K] dF^T — 8.6 KB instead of two full operand-order copies (16.9 KB); the difference holds the producer's block descriptors.
__device__ __forceinline__ bool in_K(int c) { return (c >= 3 && c <= 8) || (c >= 21 && c <= 30); }
__device__ __forceinline__ void store_rows_K(double *XK, const mfma_d4 acc[4]) {   // XK[kidx(row)][col] = X[row][col], row in K
  const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * (t >> 1) + lk + 4 * r;
      if (in_K(row)) XK[pb::kidx(row) * FLD + 16 * (t & 1) + lr] = acc[t][r];
    }
}
__device__ __forceinline__ void store_cols_K(double *XK, const mfma_d4 acc[4]) {   // XK[row][kidx(col)] = X[row][col], col in K
  const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = 16 * (t & 1) + lr;
    if (in_K(col)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) XK[(16 * (t >> 1) + lk + 4 * r) * FCLD + pb::kidx(col)] = acc[t][r];
    }
  }
}
constexpr int JK_N = 16 * FLD, PQ_N = 32 * FCLD;   // PQ: P[K, :] (16 x FLD = 528) and Q[:, K] (32 x FCLD = 544) take turns in one region
static_assert(PQ_N >= 16 * FLD, "the region holds either");
__device__ __forceinline__ void jac_cov_update_regs(const FvOperands &o, double *JK, double *PQ, mfma_d4 accJ[4], mfma_d4 accP[4]) {
  const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
  auto fk = [&](const double *BK, mfma_d4 acc[4]) {   // acc += dFc * X[K, :] (BK: the rows K of X, compact order); operands in flight first
    double b0[4], b1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { const int kc = 4 * kk + lk; b0[kk] = BK[kc * FLD + lr]; b1[kk] = BK[kc * FLD + 16 + lr]; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bool hi = !((pb::DF_KSTEPS_ROWS_LO_ONLY >> kk) & 1u);
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.f0[kk], b0[kk], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.f0[kk], b1[kk], acc[1], 0, 0, 0);
      if (hi) {
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.f1[kk], b0[kk], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.f1[kk], b1[kk], acc[3], 0, 0, 0);
      }
    }
  };
  fk(JK, accJ);   // J + dF J
  fk(PQ, accP);   // Q = P + dF P
  wave_fence();
  store_rows_K(JK, accJ);
  store_cols_K(PQ, accP);   // Q[:, K]
  wave_fence();
  {   // Q + Q[:, K] dFc^T
    double a0[4], a1[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { const int kc = 4 * kk + lk; a0[kk] = PQ[lr * FCLD + kc]; a1[kk] = PQ[(16 + lr) * FCLD + kc]; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bool hi = !((pb::DF_KSTEPS_ROWS_LO_ONLY >> kk) & 1u);
      accP[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], o.f0[kk], accP[0], 0, 0, 0);
      if (hi) accP[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], o.f1[kk], accP[1], 0, 0, 0);
      accP[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], o.f0[kk], accP[2], 0, 0, 0);
      if (hi) accP[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], o.f1[kk], accP[3], 0, 0, 0);
    }
  }
#pragma unroll
  for (int kk = 0; kk < 12; ++kk) {   // + V N V^T
    const bool lo = !((pb::V_KSTEPS_ROWS_HI_ONLY >> kk) & 1u), hi = !((pb::V_KSTEPS_ROWS_LO_ONLY >> kk) & 1u);
    const double a0 = o.v0[kk] * o.sc[kk], a1 = o.v1[kk] * o.sc[kk];
    if (lo) accP[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, o.v0[kk], accP[0], 0, 0, 0);
    if (lo && hi) {
      accP[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, o.v1[kk], accP[1], 0, 0, 0);
      accP[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, o.v0[kk], accP[2], 0, 0, 0);
    }
    if (hi) accP[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, o.v1[kk], accP[3], 0, 0, 0);
  }
  wave_fence();
  store_rows_K(PQ, accP);   // P[K, :] of the next step
  wave_fence();
}

__device__ inline void put33(double *M, int ld, int r0, int c0, const m3 &A) {
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) M[(r0 + a) * ld + c0 + b] = A.a[3 * a + b];
}

constexpr int LT_N = pb::REC_N;   // doubles of the record of one (sample, leg)
static_assert(LT_N == VILO_LEG_REC, "record size");

// The part of the leg terms of one (sample, leg) that depends on the sample and the linearisation point only (:232-287 without the
// rotation of the integration state): f, J, v, and g, h before their rotation R_e (g = -(R_e g0), h = R_e h0). A sample is the second
// endpoint of one step and the first of the next, and the forward kinematics with its derivatives (six sin / cos) is the longest
// stretch of a step, so an interval evaluates these records for all its samples up front with lane = (sample, leg) and parks them in HBM,
// laid out as the step's matrix pool wants them (preint_blocks.hpp): [v]x, [p_br + R_br f]x, h_0, J (3 x 3 each), g_0, v.
__device__ __forceinline__ void leg_sample_terms(const vilo_config &cfg, const vilo_sample &ss, int j, double rho_j, const v3 &bg, const m3 &Rbr, const v3 &pbr,
                                                 double *out /* LT_N */) {
  LegKin k;
  leg_kin_full(ss.phi + 3 * j, rho_j, cfg.rho_fix[j], k);
  const v3 dphi = ld3(ss.dphi + 3 * j);
  const m3 Rw = skew(ld3(ss.gyr) - bg);
  const v3 v = -(Rbr * (k.J * dphi)) - Rw * (pbr + Rbr * k.f);
  // (dphi^T kron I) dJ/drho = dJ_drho * dphi ; (dphi^T kron I) dJ/dq = [dJ0 dphi, dJ1 dphi, dJ2 dphi]
  const v3 g0 = Rbr * (k.dJ_drho * dphi) + Rw * (Rbr * k.df_drho);
  m3 K;
  const v3 k0 = k.dJ[0] * dphi, k1 = k.dJ[1] * dphi, k2 = k.dJ[2] * dphi;
  K.a[0] = k0.x; K.a[3] = k0.y; K.a[6] = k0.z;
  K.a[1] = k1.x; K.a[4] = k1.y; K.a[7] = k1.z;
  K.a[2] = k2.x; K.a[5] = k2.y; K.a[8] = k2.z;
  const m3 h0 = Rbr * K + Rw * Rbr * k.J;
  const m3 skv = skew(v), skp = skew(pbr + Rbr * k.f);
  for (int q = 0; q < 9; ++q) { out[q] = skv.a[q]; out[9 + q] = skp.a[q]; out[18 + q] = h0.a[q]; out[27 + q] = k.J.a[q]; }
  st3(out + 36, g0); st3(out + 39, v);
}


}  // namespace

// One IMULegIntegrationBase, batch form (STREAM = false: constructor + every push_back of the interval, state starts at the
// identity) or streaming form (STREAM = true: the object lives in HBM between calls, this call push_back()s the new samples).
// Both run the same arithmetic in the same order: pushing an interval in pieces gives bitwise the batch result.
//
// A workgroup is a PAIR of waves: the producer (wave 0) carries the state and builds each step's dF, V and noise diagonal in LDS (FP64
// VALU and LDS work, ~ 9 k cycles a step); the consumer (wave 1) takes its matrix-instruction operands of that step out of LDS into registers
// in one go, hands the block back and runs the 78 matrix instructions of jacobian <- F jacobian, covariance <- F covariance F^T + V N V^T
// (~ 5 k cycles of the matrix pipe) while the producer is already on the next step. One s_barrier per step ("blocks ready"); "blocks taken"
// is an LDS flag the producer finds set long before it needs it. Neither role needs more than half the register file, so the two waves
// of four intervals share a CU's four SIMDs as before (LDS: four per CU) and a step costs max(producer, consumer) instead of the sum.
template <bool STREAM>
__device__ __forceinline__ void preint_imu_leg_body(const vilo_config &cfg, const vilo_sample *samples, int s_begin, int s_end, const double *ln, vilo_preint *outp,
                                    PreintStream *st, double *terms /* HBM scratch of this interval: 4 * LT_N doubles per sample (+ one sample when STREAM) */,
                                    double *ff_io = nullptr /* repropagate(): the object's force filter (36 doubles, O_FF layout), in and out */) {
  // dF (its 16 non-zero columns), V, the step's 3 x 3 matrix pool, coefficients and noise diagonal: one array, the offsets of
  // preint_blocks.hpp. Padded to 32 rows (48 noise columns) with odd leading dimensions: row 31 and noise 46, 47 stay zero, so the FP64
  // MFMA tiles of jac_cov_update_regs need no masks.
  __shared__ double Ls[pb::PB_TOTAL];
  __shared__ double JK[JK_N], PQ[PQ_N];   // the consumer's (jac_cov_update_regs)
  __shared__ int fv_taken;                // steps whose operands the consumer has in registers
  __shared__ int swap_roles;
  __shared__ int pool_ready;              // steps whose 3 x 3 matrix pool (products included) and coefficients are complete
  // the producer's descriptors of the lane-parallel block construction: product / block (8 round + lane / 9), entry lane % 9 (lane 63: the
  // no-op). In LDS, [round][lane]: as registers they would put the producer over the half register file it shares a SIMD in.
  __shared__ unsigned pdesc[pb::N_PROD_ROUNDS * 64];
  __shared__ unsigned long long bdesc[pb::N_BLK_ROUNDS * 64];
  double *const Fm = Ls + pb::O_FC, *const Vm = Ls + pb::O_VM, *const nd = Ls + pb::O_ND;
  const int tid = threadIdx.x, lane = tid & 63;
  const int pb_grp = lane / 9, pb_ent = lane - 9 * pb_grp;
  if (tid < 64) {
    for (int r = 0; r < pb::N_PROD_ROUNDS; ++r) pdesc[64 * r + lane] = c_pb_tab.prod[8 * r + pb_grp];
    for (int r = 0; r < pb::N_BLK_ROUNDS; ++r) bdesc[64 * r + lane] = c_pb_tab.blk[8 * r + pb_grp];
  }
  if (STREAM) ln = st->rec.lin_ba;   // lin_ba(3) lin_bg(3) lin_rho(4) are consecutive in vilo_preint
  const v3 ba = ld3(ln), bg = ld3(ln + 3);
  double rho[4] = {ln[6], ln[7], ln[8], ln[9]};
  // state (uniform; kept redundantly in every lane's registers, updated identically)
  v3 dp = mk3(0, 0, 0), dv = mk3(0, 0, 0);
  quat dq = mkq(1, 0, 0, 0);
  v3 eps[4];
  for (int j = 0; j < 4; ++j) eps[j] = mk3(0, 0, 0);
  double sum_dt = 0.0;
  // type-2 contact filter state (imu_leg_integration_base.h:100-108) lives in LDS (zeroed with the rest below): min [0..4), max [4..8),
  // variance [8..12), window [12 + 5 leg + k], index [32..36). Every lane reads it (broadcast), lane 0 writes it back.
  double *const ffs = Ls + pb::O_FF;
  if (STREAM) {
    const vilo_preint &r = st->rec;
    dp = ld3(r.delta_p); dv = ld3(r.delta_v);
    dq = mkq(r.delta_q[3], r.delta_q[0], r.delta_q[1], r.delta_q[2]);
    for (int j = 0; j < 4; ++j) eps[j] = ld3(r.delta_eps + 3 * j);
    sum_dt = r.sum_dt;
  }
  // dF = F - I and V have a fixed sparsity pattern: zeroed once, every sample overwrites the same entries
  for (int e = tid; e < pb::PB_TOTAL; e += 128) Ls[e] = 0.0;
  if (tid == 0) {
    fv_taken = 0;
    pool_ready = 0;
    // Which wave produces: a SIMD hosts two waves of two pairs, and it should get one producer (FP64 VALU) and one consumer (matrix pipe)
    // rather than two of a kind. The pairs that share SIMDs differ in the wave slot they were given there (HW_ID bits 3:0), so the parity
    // of wave 0's slot picks its role; any choice computes the same record.
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    swap_roles = (int)(hw_id & 1u);
  }
  __syncthreads();
  const bool producer = __builtin_amdgcn_readfirstlane((tid >> 6) ^ swap_roles) == 0;
  const m3 Rbr = ld_m3_rowmajor(cfg.R_br);
  const v3 pbr = ld3(cfg.p_br);
  const m3 I3 = m3_eye();
  if (tid < 9) { Ls[pb::O_POOL + 9 * pb::S_RBR + lane] = cfg.R_br[lane]; Ls[pb::O_POOL + 9 * pb::S_I + lane] = (lane % 4 == 0) ? 1.0 : 0.0; }
  if (tid == 0) {
    const double an2 = cfg.acc_n * cfg.acc_n, anz2 = cfg.acc_n_z * cfg.acc_n_z, gn2 = cfg.gyr_n * cfg.gyr_n;
    const double aw2 = cfg.acc_w * cfg.acc_w, gw2 = cfg.gyr_w * cfg.gyr_w, pn2 = cfg.phi_n * cfg.phi_n, dpn2 = cfg.dphi_n * cfg.dphi_n;
    nd[0] = an2; nd[1] = an2; nd[2] = anz2; nd[3] = gn2; nd[4] = gn2; nd[5] = gn2;
    nd[6] = an2; nd[7] = an2; nd[8] = anz2; nd[9] = gn2; nd[10] = gn2; nd[11] = gn2;
    for (int k = 0; k < 3; ++k) { nd[12 + k] = aw2; nd[15 + k] = gw2; }
    for (int k = 0; k < 6; ++k) { nd[18 + k] = pn2; nd[24 + k] = dpn2; }
  }
  // IMULegIntegrationBase::repropagate (imu_leg_integration_base.cpp:62-86) resets what the constructor sets EXCEPT the contact-force filter
  // (foot_force_min / max / window / window_idx / var, :29-41): a re-integration starts from the filter state the previous pass over the
  // samples left in the object (contact_sensor_type 2 only: the other types have no filter)
  if (!STREAM && ff_io && cfg.contact_sensor_type == 2 && tid < 36) ffs[lane] = ff_io[lane];
  if (STREAM && tid < 36) ffs[lane] = lane < 4 ? st->ff_min[lane] : lane < 8 ? st->ff_max[lane - 4] : lane < 12 ? st->ff_var[lane - 8] : lane < 32 ? st->ff_win[lane - 12] : (double)st->ff_idx[lane - 32];
  // leg terms of every sample the steps below touch, lane = (sample, leg); slot 0 = the sample before the first step
  // (batch: the constructor's sample; streaming: the last sample of the previous push)
  {
    const int nslots = s_end - s_begin + (STREAM ? 1 : 0);
    for (int idx = tid; idx < 4 * nslots; idx += 128) {
      const int slot = idx >> 2, j = idx & 3;
      const vilo_sample &ss = (STREAM && slot == 0) ? st->last : samples[s_begin + slot - (STREAM ? 1 : 0)];
      const double rho_j = j == 0 ? rho[0] : (j == 1 ? rho[1] : (j == 2 ? rho[2] : rho[3]));   // (a lane-indexed register array would go to scratch)
      leg_sample_terms(cfg, ss, j, rho_j, bg, Rbr, pbr, terms + (size_t)idx * LT_N);
    }
    __syncthreads();
  }

  vilo_preint &o = *outp;
  const int si_first = STREAM ? s_begin : s_begin + 1;
  if (!producer) {
    // jacobian and covariance in accumulator order (register r of tile t of lane (lr, lk): row 16 (t / 2) + lk + 4 r, column 16 (t % 2) + lr):
    // the object's own (streaming), I and 0 (a fresh integration); row / column 31 are padding
    mfma_d4 accJ[4], accP[4];
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * (t >> 1) + lk + 4 * r, col = 16 * (t & 1) + lr;
        const bool in = row < 31 && col < 31;
        if (STREAM) { accJ[t][r] = in ? st->rec.jacobian[row * 31 + col] : 0.0; accP[t][r] = in ? st->rec.covariance[row * 31 + col] : 0.0; }
        else { accJ[t][r] = (in && row == col) ? 1.0 : 0.0; accP[t][r] = 0.0; }
      }
    store_rows_K(JK, accJ);
    store_rows_K(PQ, accP);
    int n_taken = 0;
    for (int si = si_first; si < s_end; ++si) {
      // its share of the step's blocks (the producer builds the others meanwhile), once the pool they are sums of is complete
      const double dt = samples[si].dt;
      while (__hip_atomic_load(&pool_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= n_taken) __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int r = BLK_ROUNDS_PRODUCER; r < pb::N_BLK_ROUNDS; ++r) pb::block_entry(bdesc[64 * r + lane], pb_ent, Ls);
      pb::tail_entry(lane, dt, Ls);
      pair_barrier();   // this step's dF, V, nd are in LDS
      FvOperands op;
      load_fv_operands(Fm, Vm, nd, op);
      __hip_atomic_store(&fv_taken, ++n_taken, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // (behind the loads: LDS is in order)
      jac_cov_update_regs(op, JK, PQ, accJ, accP);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * (t >> 1) + lk + 4 * r, col = 16 * (t & 1) + lr;
        if (row < 31 && col < 31) { o.jacobian[row * 31 + col] = accJ[t][r]; o.covariance[row * 31 + col] = accP[t][r]; }
      }
    return;
  }
  // ---- the producer ----
  // the configuration's scalars the steps read, in scalar registers (behind the fences of the loop they would be fetched again every step,
  // one exposed round trip after the other)
  const int c_contact_sensor_type = __builtin_amdgcn_readfirstlane(cfg.contact_sensor_type);
  const double c_v_n_force_thres_ratio = uni_d(cfg.v_n_force_thres_ratio), c_v_n_term1_steep = uni_d(cfg.v_n_term1_steep), c_v_n_max = uni_d(cfg.v_n_max),
               c_v_n_min_xy = uni_d(cfg.v_n_min_xy), c_v_n_min_z = uni_d(cfg.v_n_min_z), c_v_n_min = uni_d(cfg.v_n_min),
               c_v_n_term2_var_rescale = uni_d(cfg.v_n_term2_var_rescale), c_v_n_term3_distance_rescale = uni_d(cfg.v_n_term3_distance_rescale),
               c_rho_c_n = uni_d(cfg.rho_c_n), c_rho_nc_n = uni_d(cfg.rho_nc_n);
  // the records of a step's two samples, one step ahead of their use: element lane + 64 q of [endpoint][leg][LT_N]
  double rec[6];
  int rec_dst[6], rec_src[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int idx = min(lane + 64 * q, 8 * LT_N - 1);
    const int e = idx >= 4 * LT_N ? 1 : 0, r4 = idx - 4 * LT_N * e, j = r4 / LT_N, r = r4 - LT_N * j;
    rec_dst[q] = pb::record_dest(j, e, r);
    rec_src[q] = e * (4 * LT_N) + r4;
    rec[q] = 0.0;
  }
  auto load_records = [&](int si) {
    const double *src = terms + (size_t)(si - 1 - s_begin + (STREAM ? 1 : 0)) * (4 * LT_N);
#pragma unroll
    for (int q = 0; q < 6; ++q) rec[q] = src[rec_src[q]];
  };
  if (si_first < s_end) load_records(si_first);
  // the IMU rows and contact values of a step's two samples: the first one's carried over from the previous step in scalar registers, the
  // second one's fetched one step ahead (acc 0 .. 2, gyr 3 .. 5, c 6 .. 9, dt 10)
  double smp0[10], smp_next[11];
  auto fetch_sample = [&](const vilo_sample &sm, double *v) {
    for (int k = 0; k < 3; ++k) { v[k] = sm.acc[k]; v[3 + k] = sm.gyr[k]; }
    for (int k = 0; k < 4; ++k) v[6 + k] = sm.c[k];
  };
  {
    double v[10];
    fetch_sample(STREAM ? st->last : samples[s_begin], v);
    for (int k = 0; k < 10; ++k) smp0[k] = uni_d(v[k]);
  }
  if (si_first < s_end) { fetch_sample(samples[si_first], smp_next); smp_next[10] = samples[si_first].dt; }
  int n_built = 0;
  for (int si = si_first; si < s_end; ++si) {
    double smp1[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) smp1[k] = uni_d(smp_next[k]);
    if (si + 1 < s_end) { fetch_sample(samples[si + 1], smp_next); smp_next[10] = samples[si + 1].dt; }
    const double dt = smp1[10];
    const v3 acc_0 = ld3(smp0), gyr_0 = ld3(smp0 + 3), acc_1 = ld3(smp1), gyr_1 = ld3(smp1 + 3);
    // IMU midpoint update (:152-160)
    const v3 un_acc_0 = qrot(dq, acc_0 - ba);
    const v3 un_gyr = (gyr_0 + gyr_1) * 0.5 - bg;
    const quat rq = qmul(dq, mkq(1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2));
    const v3 un_acc_1 = qrot(rq, acc_1 - ba);
    const v3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
    const v3 r_dp = dp + dv * dt + un_acc * (0.5 * dt * dt);
    const v3 r_dv = dv + un_acc * dt;
    const v3 w0 = gyr_0 - bg, w1 = gyr_1 - bg;
    const m3 R0 = qR(dq), R1 = qR(rq);
    // contact flags (:183-229); integer-valued as in the reference (Vector4i foot_contact_flag)
    int flag[4];
    double ff_var[4] = {0.0, 0.0, 0.0, 0.0};
    if (c_contact_sensor_type == 0 || c_contact_sensor_type == 1) {
      for (int j = 0; j < 4; ++j) flag[j] = smp1[6 + j] >= 0.5 ? 1 : 0;
    } else {
      for (int j = 0; j < 4; ++j) {
        const double force_mag = 0.5 * (smp0[6 + j] + smp1[6 + j]);
        double fmn = ffs[j], fmx = ffs[4 + j];
        if (force_mag < fmn) fmn = 0.9 * fmn + 0.1 * force_mag;
        if (force_mag > fmx) fmx = 0.9 * fmx + 0.1 * force_mag;
        fmn *= 0.9991;
        fmx *= 0.997;
        const double thr = fmn + c_v_n_force_thres_ratio * (fmx - fmn);
        flag[j] = (int)(1.0 / (1 + exp(-c_v_n_term1_steep * (force_mag - thr))));
        const int fidx = ((int)ffs[32 + j] + 1) % 5;
        double win[5];
        for (int k = 0; k < 5; ++k) win[k] = (k == fidx) ? force_mag : ffs[12 + 5 * j + k];
        double mean = 0;
        for (int k = 0; k < 5; ++k) mean += win[k];
        mean /= 5;
        double ss = 0;
        for (int k = 0; k < 5; ++k) ss += (win[k] - mean) * (win[k] - mean);
        ff_var[j] = ss / 4;
        // (LDS operations of one wave complete in order: the reads above precede these writes)
        if (lane == 0) { ffs[j] = fmn; ffs[4 + j] = fmx; ffs[8 + j] = ff_var[j]; ffs[12 + 5 * j + fidx] = force_mag; ffs[32 + j] = (double)fidx; }
      }
    }
    // dF = F - I and V (:232-287, :376-465), every 3 x 3 product entry and every output entry on its own lane (preint_blocks.hpp).
    // The pool's inputs: the two samples' records (336 doubles, six coalesced loads, issued one step ahead) go to the slots of
    // (leg, endpoint); lane 0 adds the step's own matrices and the powers of dt.
    const v3 a0 = acc_0 - ba, a1 = acc_1 - ba;
    {
#pragma unroll
      for (int q = 0; q < 6; ++q)
        if (lane + 64 * q < 8 * LT_N) Ls[rec_dst[q]] = rec[q];
      if (si + 1 < s_end) load_records(si + 1);
      if (lane == 0) {
        const m3 Rwx = skew(un_gyr), Ra0 = skew(a0), Ra1 = skew(a1);
        const m3 kappa_7 = I3 - Rwx * dt;
        double *pool = Ls + pb::O_POOL;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          pool[9 * pb::S_R0 + q] = R0.a[q]; pool[9 * pb::S_R1 + q] = R1.a[q]; pool[9 * pb::S_K7 + q] = kappa_7.a[q];
          pool[9 * pb::S_RA0 + q] = Ra0.a[q]; pool[9 * pb::S_RA1 + q] = Ra1.a[q]; pool[9 * pb::S_RWX + q] = Rwx.a[q];
        }
        pb::coefficients(dt, Ls + pb::O_COEF);
      }
    }
    wave_fence();
    // first-level products R_e X and -(R_e g_0)
#pragma unroll
    for (int r = 0; r < 4; ++r) pb::product_entry(pdesc[64 * r + lane], pb_ent, Ls);
    pb::gvec_entry(lane, Ls);
    // the halves q_e v of the leg-odometry velocities (:245): lane 2 leg + endpoint, lanes 0 .. 7
    {
      const int q = lane & 7;
      const quat Q = (q & 1) ? rq : dq;
      const v3 r = qrot(Q, ld3(Ls + pb::O_VV + 3 * q));
      if (lane < 8) st3(Ls + pb::O_LOV + 3 * q, r);
    }
    wave_fence();
    // second level: (R_1 [a_1]x) kappa_7, (R_1 [v_1]x) kappa_7, (R_e R_br) J_e
#pragma unroll
    for (int r = 4; r < pb::N_PROD_ROUNDS; ++r) pb::product_entry(pdesc[64 * r + lane], pb_ent, Ls);
    __hip_atomic_store(&pool_ready, n_built + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // the consumer starts on its blocks
    // epsilon update + noise (uniform, every lane) (:245, :288-374)
    v3 lo_v[4], r_eps[4];
    for (int j = 0; j < 4; ++j) {
      lo_v[j] = (ld3(Ls + pb::O_LOV + 6 * j) + ld3(Ls + pb::O_LOV + 6 * j + 3)) * 0.5;
      r_eps[j] = eps[j] + lo_v[j] * dt;
    }
    // nd, dF and V of the previous step: the consumer has them in registers (set ~ a step ago)
    while (__hip_atomic_load(&fv_taken, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < n_built) __builtin_amdgcn_s_sleep(1);
    if (lane == 0) {
      double unc[12], rho_unc[4];
      if (c_contact_sensor_type == 0 || c_contact_sensor_type == 1) {
        for (int j = 0; j < 4; ++j) {
          const double n_xy = c_v_n_max * (1 - flag[j]) + flag[j] * c_v_n_min_xy;
          const double n_z = c_v_n_max * (1 - flag[j]) + flag[j] * c_v_n_min_z;
          unc[3 * j] = n_xy; unc[3 * j + 1] = n_xy; unc[3 * j + 2] = n_z;
        }
      } else {
        for (int j = 0; j < 4; ++j) {
          const double n1 = c_v_n_max * (1 - flag[j]) + c_v_n_min;
          const double n2 = c_v_n_term2_var_rescale * ff_var[j];
          const v3 tmp = lo_v[j] - dv;
          unc[3 * j] = n1 + n2 + c_v_n_term3_distance_rescale * tmp.x * tmp.x;
          unc[3 * j + 1] = n1 + n2 + c_v_n_term3_distance_rescale * tmp.y * tmp.y;
          unc[3 * j + 2] = n1 + n2 + c_v_n_term3_distance_rescale * tmp.z * tmp.z;
        }
      }
      int fsum = 0;
      for (int j = 0; j < 4; ++j) { rho_unc[j] = c_rho_c_n * flag[j] + c_rho_nc_n; fsum += flag[j]; }
      if (fsum < 1e-6) {
        for (int j = 0; j < 4; ++j) rho_unc[j] = c_rho_nc_n;
        for (int k = 0; k < 12; ++k) unc[k] = 10e10;
      }
      // (nd[0 .. 30): the IMU, bias and encoder variances do not change from step to step: written before the loop)
      for (int k = 0; k < 12; ++k) nd[30 + k] = unc[k];
      for (int k = 0; k < 4; ++k) nd[42 + k] = rho_unc[k];
    }
    wave_fence();
#pragma unroll
    for (int r = 0; r < BLK_ROUNDS_PRODUCER; ++r) pb::block_entry(bdesc[64 * r + lane], pb_ent, Ls);
    pair_barrier();   // blocks ready: the consumer takes them
    ++n_built;
#pragma unroll
    for (int k = 0; k < 10; ++k) smp0[k] = smp1[k];
    // propagate() (:88-136)
    dp = r_dp; dv = r_dv; dq = qnormalized(rq);
    for (int j = 0; j < 4; ++j) eps[j] = r_eps[j];
    sum_dt += dt;
  }
  if (!STREAM && ff_io && cfg.contact_sensor_type == 2) {
    wave_fence();   // (lane 0's last filter update)
    if (lane < 36) ff_io[lane] = ffs[lane];
  }
  if (lane == 0) {
    o.sum_dt = sum_dt;
    st3(o.delta_p, dp); st3(o.delta_v, dv);
    o.delta_q[0] = dq.x; o.delta_q[1] = dq.y; o.delta_q[2] = dq.z; o.delta_q[3] = dq.w;
    for (int j = 0; j < 4; ++j) { st3(o.delta_eps + 3 * j, eps[j]); o.lin_rho[j] = rho[j]; }
    st3(o.lin_ba, ba); st3(o.lin_bg, bg);
    if (STREAM && s_end > s_begin) {
      st->last = samples[s_end - 1];
      st->n_pushed += s_end - s_begin;
      for (int j = 0; j < 4; ++j) {
        st->ff_min[j] = ffs[j]; st->ff_max[j] = ffs[4 + j]; st->ff_var[j] = ffs[8 + j]; st->ff_idx[j] = (int)ffs[32 + j];
        for (int k = 0; k < 5; ++k) st->ff_win[5 * j + k] = ffs[12 + 5 * j + k];
      }
    }
  }
}

__global__ void __launch_bounds__(PAIR_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) k_preint_imu_leg(int n, const vilo_config *cfgp, const vilo_sample *samples, const int *offsets,
                                                       const double *lin, vilo_preint *out, double *terms) {
  const int f = blockIdx.x;
  if (f >= n) return;
  preint_imu_leg_body<false>(*cfgp, samples, offsets[f], offsets[f + 1], lin + 10 * f, out + f, nullptr, terms + (size_t)offsets[f] * (4 * LT_N));
}

// IMULegIntegrationBase::repropagate (imu_leg_integration_base.cpp:62-86) for every live interval of a resident batch, at the biases of
// the candidate state the next linearisation pass evaluates: the record an interval's factor reads is integrated again from its samples
// with linearized_ba / bg / rho = the candidate's (BASELINE configs[2]: "K1 re-propagation of all 10 intervals inside the iteration").
__global__ void __launch_bounds__(PAIR_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) k_repropagate(BatchDev b, const vilo_config *cfgp, int mode) {
  __shared__ double lin_s[10];
  __shared__ int same_s;
  const int f = blockIdx.x, win = f / 10, k = f % 10;
  if (b.imu_skip[f] || b.rp_offsets[f + 1] <= b.rp_offsets[f]) return;
  if (mode == 1 && b.st[win].done) return;
  const double *xs = (mode ? b.xc : b.x) + (size_t)win * XSTRIDE;
  vilo_preint *rec = (vilo_preint *)b.rp_pre + f;
  if (threadIdx.x == 0) same_s = 1;
  __syncthreads();
  if (threadIdx.x < 10) {
    // (mode 2: the object's original integration, at the record's own linearisation point — vilo_batch_set_samples runs it once for
    // contact_sensor_type 2 so that the force filter holds what constructor + push_back left when the first repropagate() comes)
    const double v = mode == 2 ? rec->lin_ba[threadIdx.x] : threadIdx.x < 6 ? xs[XO_SB + 9 * k + 3 + threadIdx.x] : xs[XO_LB + 4 * k + (threadIdx.x - 6)];
    lin_s[threadIdx.x] = v;
    if (v != rec->lin_ba[threadIdx.x]) same_s = 0;   // lin_ba(3) lin_bg(3) lin_rho(4) are consecutive in vilo_preint
  }
  __syncthreads();
  if (!mode && same_s) return;   // (marginalisation) the record already is the one integrated at the accepted state
  preint_imu_leg_body<false>(*cfgp, b.rp_samples, b.rp_offsets[f], b.rp_offsets[f + 1], lin_s, rec, nullptr, b.rp_terms + (size_t)b.rp_offsets[f] * (4 * LT_N),
                             b.rp_ff ? b.rp_ff + (size_t)f * VILO_FF_N : nullptr);
}

// push_back() on device-resident objects: workgroup k appends samples[offsets[k] .. offsets[k+1]) to stream ids[k]
__global__ void __launch_bounds__(PAIR_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) k_preint_stream_push(int n, const vilo_config *cfgp, const vilo_sample *samples, const int *offsets, const int *ids,
                                                           PreintStream *streams, double *terms) {
  const int f = blockIdx.x;
  if (f >= n) return;
  PreintStream *st = streams + ids[f];
  preint_imu_leg_body<true>(*cfgp, samples, offsets[f], offsets[f + 1], nullptr, &st->rec, st, terms + (size_t)(offsets[f] + f) * (4 * LT_N));
}

// IMULegIntegrationBase{acc_0, gyr_0, phi_0, dphi_0, c_0, ba, bg, rho} (imu_leg_integration_base.cpp:7-42)
__global__ void __launch_bounds__(64) k_preint_stream_reset(int n, const int *ids, const vilo_sample *first, const double *lin, PreintStream *streams) {
  const int f = blockIdx.x, lane = threadIdx.x;
  if (f >= n) return;
  PreintStream &st = streams[ids[f]];
  vilo_preint &r = st.rec;
  for (int e = lane; e < 31 * 31; e += 64) { r.jacobian[e] = (e / 31 == e % 31) ? 1.0 : 0.0; r.covariance[e] = 0.0; }
  if (lane == 0) {
    r.sum_dt = 0.0;
    for (int k = 0; k < 3; ++k) { r.delta_p[k] = 0.0; r.delta_v[k] = 0.0; r.delta_q[k] = 0.0; r.lin_ba[k] = lin[10 * f + k]; r.lin_bg[k] = lin[10 * f + 3 + k]; }
    r.delta_q[3] = 1.0;
    for (int k = 0; k < 12; ++k) r.delta_eps[k] = 0.0;
    for (int k = 0; k < 4; ++k) { r.lin_rho[k] = lin[10 * f + 6 + k]; st.ff_min[k] = st.ff_max[k] = st.ff_var[k] = 0.0; st.ff_idx[k] = 0; }
    for (int k = 0; k < 20; ++k) st.ff_win[k] = 0.0;
    st.last = first[f];
    st.n_pushed = 0;
  }
}

__global__ void k_preint_stream_gather(int n, const int *ids, const int *dst_idx /* null: f */, const PreintStream *streams, vilo_preint *out) {
  const int f = blockIdx.x;
  if (f >= n) return;
  const double *src = (const double *)&streams[ids[f]].rec;
  double *dst = (double *)(out + (dst_idx ? dst_idx[f] : f));
  for (int e = threadIdx.x; e < (int)(sizeof(vilo_preint) / sizeof(double)); e += blockDim.x) dst[e] = src[e];
}

template <bool STREAM>
__device__ __forceinline__ void preint_imu_body(const vilo_config &cfg, const vilo_sample *samples, int s_begin, int s_end, const double *ln, vilo_preint_imu *outp,
                                PreintImuStream *st) {
  __shared__ double Fm[15 * 15], Vm[15 * 18], nd[18];
  __shared__ double Jm[15 * 16], Pm[15 * 16], Qm[15 * 16];
  const int lane = threadIdx.x;
  if (STREAM) ln = st->rec.lin_ba;   // lin_ba(3) lin_bg(3) are consecutive in vilo_preint_imu
  const v3 ba = ld3(ln), bg = ld3(ln + 3);
  v3 dp = mk3(0, 0, 0), dv = mk3(0, 0, 0);
  quat dq = mkq(1, 0, 0, 0);
  double sum_dt = 0.0;
  if (STREAM) {
    const vilo_preint_imu &r = st->rec;
    dp = ld3(r.delta_p); dv = ld3(r.delta_v);
    dq = mkq(r.delta_q[3], r.delta_q[0], r.delta_q[1], r.delta_q[2]);
    sum_dt = r.sum_dt;
    for (int e = lane; e < 15 * 16; e += 64) { Jm[e] = 0.0; Pm[e] = 0.0; }
    __syncthreads();
    for (int e = lane; e < 15 * 15; e += 64) { Jm[(e / 15) * 16 + (e % 15)] = r.jacobian[e]; Pm[(e / 15) * 16 + (e % 15)] = r.covariance[e]; }
  } else {
    for (int e = lane; e < 15 * 16; e += 64) { Jm[e] = ((e / 16) == (e % 16)) ? 1.0 : 0.0; Pm[e] = 0.0; }
  }
  if (lane < 18) {   // integration_base.h:31-37 (ACC_N on all axes)
    const int blk = lane / 3;
    nd[lane] = (blk == 0 || blk == 2) ? cfg.acc_n * cfg.acc_n : (blk == 1 || blk == 3) ? cfg.gyr_n * cfg.gyr_n : (blk == 4) ? cfg.acc_w * cfg.acc_w : cfg.gyr_w * cfg.gyr_w;
  }
  __syncthreads();
  const m3 I3 = m3_eye();
  for (int si = STREAM ? s_begin : s_begin + 1; si < s_end; ++si) {
    const vilo_sample &s0 = (STREAM && si == s_begin) ? st->last : samples[si - 1], &s1 = samples[si];
    const double dt = s1.dt;
    const v3 acc_0 = ld3(s0.acc), gyr_0 = ld3(s0.gyr), acc_1 = ld3(s1.acc), gyr_1 = ld3(s1.gyr);
    const v3 un_acc_0 = qrot(dq, acc_0 - ba);
    const v3 un_gyr = (gyr_0 + gyr_1) * 0.5 - bg;
    const quat rq = qmul(dq, mkq(1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2));
    const v3 un_acc_1 = qrot(rq, acc_1 - ba);
    const v3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
    const v3 r_dp = dp + dv * dt + un_acc * (0.5 * dt * dt);
    const v3 r_dv = dv + un_acc * dt;
    for (int e = lane; e < 15 * 15; e += 64) Fm[e] = 0.0;
    for (int e = lane; e < 15 * 18; e += 64) Vm[e] = 0.0;
    __syncthreads();
    if (lane == 0) {
      const m3 R0 = qR(dq), R1 = qR(rq), Rwx = skew(un_gyr), Ra0 = skew(acc_0 - ba), Ra1 = skew(acc_1 - ba);
      const m3 K7 = I3 - Rwx * dt;
      put33(Fm, 15, 0, 0, I3);
      put33(Fm, 15, 0, 3, (R0 * Ra0) * (-0.25 * dt * dt) + (R1 * Ra1 * K7) * (-0.25 * dt * dt));
      put33(Fm, 15, 0, 6, I3 * dt);
      put33(Fm, 15, 0, 9, (R0 + R1) * (-0.25 * dt * dt));
      put33(Fm, 15, 0, 12, (R1 * Ra1) * (-0.25 * dt * dt * -dt));
      put33(Fm, 15, 3, 3, K7);
      put33(Fm, 15, 3, 12, I3 * (-dt));
      put33(Fm, 15, 6, 3, (R0 * Ra0) * (-0.5 * dt) + (R1 * Ra1 * K7) * (-0.5 * dt));
      put33(Fm, 15, 6, 6, I3);
      put33(Fm, 15, 6, 9, (R0 + R1) * (-0.5 * dt));
      put33(Fm, 15, 6, 12, (R1 * Ra1) * (-0.5 * dt * -dt));
      put33(Fm, 15, 9, 9, I3);
      put33(Fm, 15, 12, 12, I3);
      const m3 VpG = (R1 * Ra1) * (-0.25 * dt * dt * 0.5 * dt), VvG = (R1 * Ra1) * (-0.5 * dt * 0.5 * dt);
      put33(Vm, 18, 0, 0, R0 * (0.25 * dt * dt));
      put33(Vm, 18, 0, 3, VpG);
      put33(Vm, 18, 0, 6, R1 * (0.25 * dt * dt));
      put33(Vm, 18, 0, 9, VpG);
      put33(Vm, 18, 3, 3, I3 * (0.5 * dt));
      put33(Vm, 18, 3, 9, I3 * (0.5 * dt));
      put33(Vm, 18, 6, 0, R0 * (0.5 * dt));
      put33(Vm, 18, 6, 3, VvG);
      put33(Vm, 18, 6, 6, R1 * (0.5 * dt));
      put33(Vm, 18, 6, 9, VvG);
      put33(Vm, 18, 9, 12, I3 * dt);
      put33(Vm, 18, 12, 15, I3 * dt);
    }
    __syncthreads();
    jac_cov_update<15, 18>(Fm, Vm, nd, Jm, Pm, Qm);
    dp = r_dp; dv = r_dv; dq = qnormalized(rq);
    sum_dt += dt;
  }
  vilo_preint_imu &o = *outp;
  if (lane == 0) {
    o.sum_dt = sum_dt;
    st3(o.delta_p, dp); st3(o.delta_v, dv);
    o.delta_q[0] = dq.x; o.delta_q[1] = dq.y; o.delta_q[2] = dq.z; o.delta_q[3] = dq.w;
    st3(o.lin_ba, ba); st3(o.lin_bg, bg);
    if (STREAM && s_end > s_begin) { st->last = samples[s_end - 1]; st->n_pushed += s_end - s_begin; }
  }
  for (int e = lane; e < 15 * 15; e += 64) {
    o.jacobian[e] = Jm[(e / 15) * 16 + (e % 15)];
    o.covariance[e] = Pm[(e / 15) * 16 + (e % 15)];
  }
}

__global__ void __launch_bounds__(64) k_preint_imu(int n, const vilo_config *cfgp, const vilo_sample *samples, const int *offsets,
                                                   const double *lin, vilo_preint_imu *out, double * /* no leg terms */) {
  const int f = blockIdx.x;
  if (f >= n) return;
  preint_imu_body<false>(*cfgp, samples, offsets[f], offsets[f + 1], lin + 6 * f, out + f, nullptr);
}
__global__ void __launch_bounds__(64) k_preint_imu_stream_push(int n, const vilo_config *cfgp, const vilo_sample *samples, const int *offsets, const int *ids,
                                                               PreintImuStream *streams) {
  const int f = blockIdx.x;
  if (f >= n) return;
  PreintImuStream *st = streams + ids[f];
  preint_imu_body<true>(*cfgp, samples, offsets[f], offsets[f + 1], nullptr, &st->rec, st);
}
// new IntegrationBase{acc_0, gyr_0, ba, bg} (integration_base.h:18-40)
__global__ void __launch_bounds__(64) k_preint_imu_stream_reset(int n, const int *ids, const vilo_sample *first, const double *lin, PreintImuStream *streams) {
  const int f = blockIdx.x, lane = threadIdx.x;
  if (f >= n) return;
  PreintImuStream &st = streams[ids[f]];
  vilo_preint_imu &r = st.rec;
  for (int e = lane; e < 15 * 15; e += 64) { r.jacobian[e] = (e / 15 == e % 15) ? 1.0 : 0.0; r.covariance[e] = 0.0; }
  if (lane == 0) {
    r.sum_dt = 0.0;
    for (int k = 0; k < 3; ++k) { r.delta_p[k] = 0.0; r.delta_v[k] = 0.0; r.delta_q[k] = 0.0; r.lin_ba[k] = lin[6 * f + k]; r.lin_bg[k] = lin[6 * f + 3 + k]; }
    r.delta_q[3] = 1.0;
    st.last = first[f];
    st.n_pushed = 0;
  }
}
__global__ void k_preint_imu_stream_gather(int n, const int *ids, const int *dst_idx, const PreintImuStream *streams, vilo_preint_imu *out) {
  const int f = blockIdx.x;
  if (f >= n) return;
  const double *src = (const double *)&streams[ids[f]].rec;
  double *dst = (double *)(out + (dst_idx ? dst_idx[f] : f));
  for (int e = threadIdx.x; e < (int)(sizeof(vilo_preint_imu) / sizeof(double)); e += blockDim.x) dst[e] = src[e];
}

template <class OUT, class KERNEL>
static int preintegrate_impl(vilo_ctx *ctx, int n, const vilo_sample *samples, const int32_t *offsets, const double *lin, int lin_w,
                             OUT *out, KERNEL kern, size_t terms_per_sample, int threads) {
  if (!ctx || n < 0 || !samples || !offsets || !lin || !out) return VILO_ERR_BAD_ARG;
  if (n == 0) return VILO_OK;
  VILO_HIP(hipSetDevice(ctx->device));
  const int ns = offsets[n];
  for (int i = 0; i < n; ++i)
    if (offsets[i + 1] <= offsets[i]) return VILO_ERR_BAD_ARG;
  DevBuf d_s, d_o, d_l, d_out, d_t;
  VILO_HIP(d_t.alloc(sizeof(double) * terms_per_sample * (size_t)ns));
  VILO_HIP(d_s.alloc(sizeof(vilo_sample) * (size_t)ns));
  VILO_HIP(d_o.alloc(sizeof(int) * (size_t)(n + 1)));
  VILO_HIP(d_l.alloc(sizeof(double) * (size_t)lin_w * n));
  VILO_HIP(d_out.alloc(sizeof(OUT) * (size_t)n));
  VILO_HIP(hipMemcpyAsync(d_s.p, samples, sizeof(vilo_sample) * (size_t)ns, hipMemcpyHostToDevice, ctx->stream));
  VILO_HIP(hipMemcpyAsync(d_o.p, offsets, sizeof(int) * (size_t)(n + 1), hipMemcpyHostToDevice, ctx->stream));
  VILO_HIP(hipMemcpyAsync(d_l.p, lin, sizeof(double) * (size_t)lin_w * n, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(kern, dim3(n), dim3(threads), 0, ctx->stream, n, (const vilo_config *)ctx->d_cfg, d_s.as<vilo_sample>(),
                     d_o.as<int>(), d_l.as<double>(), d_out.as<OUT>(), d_t.as<double>());
  VILO_HIP(hipGetLastError());
  VILO_HIP(hipMemcpyAsync(out, d_out.p, sizeof(OUT) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  return VILO_OK;
}

extern "C" int vilo_preintegrate(vilo_ctx *ctx, int n, const vilo_sample *samples, const int32_t *offsets, const double *lin,
                                 vilo_preint *out) {
  return preintegrate_impl(ctx, n, samples, offsets, lin, 10, out, k_preint_imu_leg, 4 * LT_N, PAIR_THREADS);
}
extern "C" int vilo_preintegrate_imu(vilo_ctx *ctx, int n, const vilo_sample *samples, const int32_t *offsets, const double *lin,
                                     vilo_preint_imu *out) {
  return preintegrate_impl(ctx, n, samples, offsets, lin, 6, out, k_preint_imu, 0, 64);
}

// ---- device-resident, incrementally updated preintegration (the reference's push_back as samples arrive, estimator.cpp:619-626) ----

static int streams_create(vilo_ctx *ctx, int n, int kind, vilo_preint_streams **out) {
  vilo_preint_streams *s = new vilo_preint_streams();
  s->n = n; s->d = nullptr; s->di = nullptr; s->device = ctx->device; s->kind = kind;
  void *q = nullptr;
  const size_t bytes = (kind ? sizeof(PreintImuStream) : sizeof(PreintStream)) * (size_t)n;
  if (hipMalloc(&q, bytes) != hipSuccess || hipMemset(q, 0, bytes) != hipSuccess) {
    if (q) (void)hipFree(q);
    delete s;
    ctx->err = "vilo_preint_streams_create: allocation failed";
    return VILO_ERR_HIP;
  }
  if (kind) s->di = (PreintImuStream *)q; else s->d = (PreintStream *)q;
  *out = s;
  return VILO_OK;
}

extern "C" int vilo_preint_streams_create(vilo_ctx *ctx, int n, vilo_preint_streams **out) {
  if (!ctx || n <= 0 || !out) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  return streams_create(ctx, n, 0, out);
}
extern "C" int vilo_preint_streams_create_imu(vilo_ctx *ctx, int n, vilo_preint_streams **out) {
  if (!ctx || n <= 0 || !out) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  return streams_create(ctx, n, 1, out);
}
extern "C" void vilo_preint_streams_destroy(vilo_ctx *ctx, vilo_preint_streams *s) {
  if (!s) return;
  (void)ctx;   // may already be gone when a host object releases its pool late
  (void)hipSetDevice(s->device);
  if (s->d) (void)hipFree(s->d);
  if (s->di) (void)hipFree(s->di);
  for (auto &c : s->scr) {
    if (c.done) { if (c.pending) (void)hipEventSynchronize(c.done); (void)hipEventDestroy(c.done); }
    if (c.dev) (void)hipFree(c.dev);
    if (c.host) (void)hipHostFree(c.host);
  }
  delete s;
}
// the pool's call scratch [which] with room for `bytes`, free to be written (the previous asynchronous use has completed)
static int streams_scratch(vilo_ctx *ctx, vilo_preint_streams *s, int which, size_t bytes) {
  vilo_preint_streams::Scratch &c = s->scr[which];
  if (!c.done) VILO_HIP(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
  if (c.pending) { VILO_HIP(hipEventSynchronize(c.done)); c.pending = false; }
  if (c.bytes < bytes) {
    if (c.dev) VILO_HIP(hipFree(c.dev));
    if (c.host) VILO_HIP(hipHostFree(c.host));
    c.dev = c.host = nullptr; c.bytes = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    VILO_HIP(hipMalloc(&c.dev, want));
    VILO_HIP(hipHostMalloc(&c.host, want, hipHostMallocDefault));
    c.bytes = want;
  }
  return VILO_OK;
}
static int check_ids(const vilo_preint_streams *s, int n, const int32_t *ids) {
  for (int i = 0; i < n; ++i) {
    if (ids[i] < 0 || ids[i] >= s->n) return VILO_ERR_BAD_ARG;
    for (int j = 0; j < i; ++j)
      if (ids[j] == ids[i]) return VILO_ERR_BAD_ARG;   // two workgroups would update one object
  }
  return VILO_OK;
}
extern "C" int vilo_preint_streams_reset(vilo_ctx *ctx, vilo_preint_streams *s, int n, const int32_t *ids, const vilo_sample *first, const double *lin) {
  if (!ctx || !s || n < 0 || (n && (!ids || !first || !lin))) return VILO_ERR_BAD_ARG;
  if (n == 0) return VILO_OK;
  if (check_ids(s, n, ids) != VILO_OK) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  const int lw = s->kind ? 6 : 10;
  // [first samples | linearisation points | ids] in one buffer, one copy; asynchronous on the context's stream (the caller's arrays are
  // free on return: they were copied into the pool's page-locked mirror)
  const size_t o_f = 0, o_l = o_f + sizeof(vilo_sample) * (size_t)n, o_i = o_l + sizeof(double) * lw * (size_t)n, total = o_i + sizeof(int) * (size_t)n;
  int rc = streams_scratch(ctx, s, 0, total);
  if (rc != VILO_OK) return rc;
  vilo_preint_streams::Scratch &c = s->scr[0];
  memcpy((char *)c.host + o_f, first, sizeof(vilo_sample) * (size_t)n);
  memcpy((char *)c.host + o_l, lin, sizeof(double) * lw * (size_t)n);
  memcpy((char *)c.host + o_i, ids, sizeof(int) * (size_t)n);
  VILO_HIP(hipMemcpyAsync(c.dev, c.host, total, hipMemcpyHostToDevice, ctx->stream));
  const vilo_sample *d_f = (const vilo_sample *)((char *)c.dev + o_f);
  const double *d_l = (const double *)((char *)c.dev + o_l);
  const int *d_i = (const int *)((char *)c.dev + o_i);
  if (s->kind) hipLaunchKernelGGL(k_preint_imu_stream_reset, dim3(n), dim3(64), 0, ctx->stream, n, d_i, d_f, d_l, s->di);
  else hipLaunchKernelGGL(k_preint_stream_reset, dim3(n), dim3(64), 0, ctx->stream, n, d_i, d_f, d_l, s->d);
  VILO_HIP(hipGetLastError());
  VILO_HIP(hipEventRecord(c.done, ctx->stream));
  c.pending = true;
  return VILO_OK;
}
extern "C" int vilo_preint_streams_push(vilo_ctx *ctx, vilo_preint_streams *s, int n, const int32_t *ids, const vilo_sample *samples,
                                        const int32_t *offsets) {
  if (!ctx || !s || n < 0 || (n && (!ids || !samples || !offsets))) return VILO_ERR_BAD_ARG;
  if (n == 0) return VILO_OK;
  if (check_ids(s, n, ids) != VILO_OK) return VILO_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i)
    if (offsets[i + 1] < offsets[i]) return VILO_ERR_BAD_ARG;
  const int ns = offsets[n] - offsets[0];
  if (ns == 0) return VILO_OK;
  VILO_HIP(hipSetDevice(ctx->device));
  // [samples | offsets | ids] up in one copy, the leg-term scratch of the push kernel behind them; asynchronous on the context's stream:
  // whatever uses the objects next (a batch's gather, vilo_preint_streams_read) is queued behind the kernel
  const size_t o_s = 0, o_o = o_s + sizeof(vilo_sample) * (size_t)offsets[n], o_i = o_o + sizeof(int) * (size_t)(n + 1);
  const size_t up = o_i + sizeof(int) * (size_t)n, o_t = (up + 255) & ~(size_t)255;
  const size_t total = o_t + (s->kind ? 0 : sizeof(double) * 4 * LT_N * ((size_t)offsets[n] + (size_t)n));   // leg terms: every sample + the stream's last one
  int rc = streams_scratch(ctx, s, 1, total);
  if (rc != VILO_OK) return rc;
  vilo_preint_streams::Scratch &c = s->scr[1];
  memcpy((char *)c.host + o_s, samples, sizeof(vilo_sample) * (size_t)offsets[n]);
  memcpy((char *)c.host + o_o, offsets, sizeof(int) * (size_t)(n + 1));
  memcpy((char *)c.host + o_i, ids, sizeof(int) * (size_t)n);
  VILO_HIP(hipMemcpyAsync(c.dev, c.host, up, hipMemcpyHostToDevice, ctx->stream));
  const vilo_sample *d_s = (const vilo_sample *)((char *)c.dev + o_s);
  const int *d_o = (const int *)((char *)c.dev + o_o), *d_i = (const int *)((char *)c.dev + o_i);
  if (s->kind)
    hipLaunchKernelGGL(k_preint_imu_stream_push, dim3(n), dim3(64), 0, ctx->stream, n, (const vilo_config *)ctx->d_cfg, d_s, d_o, d_i, s->di);
  else
    hipLaunchKernelGGL(k_preint_stream_push, dim3(n), dim3(PAIR_THREADS), 0, ctx->stream, n, (const vilo_config *)ctx->d_cfg, d_s, d_o, d_i, s->d,
                       (double *)((char *)c.dev + o_t));
  VILO_HIP(hipGetLastError());
  VILO_HIP(hipEventRecord(c.done, ctx->stream));
  c.pending = true;
  return VILO_OK;
}
extern "C" int vilo_preint_streams_read_imu(vilo_ctx *ctx, vilo_preint_streams *s, int n, const int32_t *ids, vilo_preint_imu *out) {
  if (!ctx || !s || !s->kind || n < 0 || (n && (!ids || !out))) return VILO_ERR_BAD_ARG;
  if (n == 0) return VILO_OK;
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= s->n) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  DevBuf d_i, d_out;
  VILO_HIP(d_i.alloc(sizeof(int) * (size_t)n)); VILO_HIP(d_out.alloc(sizeof(vilo_preint_imu) * (size_t)n));
  VILO_HIP(hipMemcpyAsync(d_i.p, ids, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_preint_imu_stream_gather, dim3(n), dim3(256), 0, ctx->stream, n, d_i.as<int>(), (const int *)nullptr, (const PreintImuStream *)s->di, d_out.as<vilo_preint_imu>());
  VILO_HIP(hipGetLastError());
  VILO_HIP(hipMemcpyAsync(out, d_out.p, sizeof(vilo_preint_imu) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  return VILO_OK;
}
extern "C" int vilo_preint_streams_read(vilo_ctx *ctx, vilo_preint_streams *s, int n, const int32_t *ids, vilo_preint *out) {
  if (!ctx || !s || s->kind || n < 0 || (n && (!ids || !out))) return VILO_ERR_BAD_ARG;
  if (n == 0) return VILO_OK;
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= s->n) return VILO_ERR_BAD_ARG;
  VILO_HIP(hipSetDevice(ctx->device));
  DevBuf d_i, d_out;
  VILO_HIP(d_i.alloc(sizeof(int) * (size_t)n)); VILO_HIP(d_out.alloc(sizeof(vilo_preint) * (size_t)n));
  VILO_HIP(hipMemcpyAsync(d_i.p, ids, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_preint_stream_gather, dim3(n), dim3(256), 0, ctx->stream, n, d_i.as<int>(), (const int *)nullptr, (const PreintStream *)s->d, d_out.as<vilo_preint>());
  VILO_HIP(hipGetLastError());
  VILO_HIP(hipMemcpyAsync(out, d_out.p, sizeof(vilo_preint) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  VILO_HIP(hipStreamSynchronize(ctx->stream));
  return VILO_OK;
}

// records of device-resident objects straight into a batch's record array (device to device): d_ids / d_dst are device arrays
int vilo_launch_preint_gather(vilo_ctx *ctx, const vilo_preint_streams *pool, int n, const int *d_ids, const int *d_dst, void *d_out) {
  if (pool->kind) hipLaunchKernelGGL(k_preint_imu_stream_gather, dim3(n), dim3(256), 0, ctx->stream, n, d_ids, d_dst, (const PreintImuStream *)pool->di, (vilo_preint_imu *)d_out);
  else hipLaunchKernelGGL(k_preint_stream_gather, dim3(n), dim3(256), 0, ctx->stream, n, d_ids, d_dst, (const PreintStream *)pool->d, (vilo_preint *)d_out);
  VILO_HIP(hipGetLastError());
  return VILO_OK;
}

// mode 1: at the candidate the next linearisation pass evaluates; mode 0: at the accepted state (before marginalisation). The
// whitening matrices of the factors follow (sqrt_info of the new covariance).
int vilo_repropagate_launch(vilo_ctx *ctx, BatchDev &b, int mode, int stage) {
  if ((!b.rp_on && mode != 2) || !b.rp_samples || !b.leg) return VILO_OK;
  if (stage == 0) {
    hipLaunchKernelGGL(k_repropagate, dim3(b.W * 10), dim3(PAIR_THREADS), 0, ctx->stream, b, (const vilo_config *)ctx->d_cfg, mode);
    VILO_HIP(hipGetLastError());
    return VILO_OK;
  }
  // the flags belong to the records integrated for THIS point (k_accept reads them as "candidate not evaluable")
  VILO_HIP(hipMemsetAsync(b.prep_bad, 0, sizeof(int) * (size_t)b.W * 10, ctx->stream));
  return vilo_launch_prepare_preint(ctx, b.W * 10, (const vilo_preint *)b.rp_pre, b.prep, b.prep_bad, b.imu_skip, 1);
}
