// Projection factors of the fused visual linearisation (k_visual_linearize), evaluated from hoisted rotation products.
//
// The three Projection*Factor::Evaluate bodies (projectionTwoFrameOneCamFactor.cpp:43-150, projectionTwoFrameTwoCamFactor.cpp:43-166,
// projectionOneFrameTwoCamFactor.cpp:42-134) multiply the same rotation matrices for every landmark: ric^T Rj^T, ric^T Rj^T Ri,
// ric^T Rj^T Ri ric depend on the (start frame, observing frame) pair only, and the landmark enters through three points
// (pts_camera_i, pts_imu_i, pts_imu_j) and the 2 x 3 projection Jacobian `reduce`. A lane (= landmark) therefore
//   * keeps what depends on its start frame alone (pts_camera_i, pts_imu_i, the world point, 1 / lambda) across its frames,
//   * reads the pair's products from a small LDS table a few lanes of the wave build per frame (vis_build_pair_table),
//   * forms T = reduce * M (2 x 3) for each product M once and gets every Jacobian block from T by a cross product:
//       reduce * M * skew(p) = rows (t x p);  e.g. d r / d theta_i = -reduce * A Ri skew(pts_imu_i) = rows (pts_imu_i x t1).
// Identities used (exact in real arithmetic, rounding-level differences against the literal bodies, which kernels_eval.hip and the
// oracle keep):  ric^T (Rj^T Ri - I) = A Ri - ric^T;   tmp_r pts_camera_i + ric^T (Rj^T (Ri tic + Pi - Pj) - tic) = pts_camera_j;
//   tmp_r pts_i_td / lambda^2 = tmp_r pts_camera_i / lambda.
// The loss is ceres::HuberLoss (estimator.cpp:1062): rho'' <= 0 everywhere, so Corrector takes its first branch for every residual
// (corrector.cc: "rho[2] <= 0": residual_scaling = sqrt(rho'), alpha_sq_norm = 0) and the corrected block is sqrt(rho') [J | r] —
// the factor is folded into `reduce` (inliers: exactly 1).
#pragma once
#include "factors.hpp"

namespace vilo {

// window-level table (doubles): camera extrinsics as matrices
#define VW_RIC 0      // ric   (9, row-major, padded to 10)
#define VW_TIC 10     // tic   (3, padded to 4)
#define VW_RIC2 14    // ric2
#define VW_TIC2 24    // tic2
#define VW_A2 28      // ric2^T ric  (OneFrameTwoCam)
#define VW_N 38
// per (segment, frame) table (doubles; every 3 x 3 block padded to 10 so that blocks start 16-byte aligned)
#define VT_RJ 0       // Rj
#define VT_PJ 10      // Pj
#define VT_A0 14      // ric^T Rj^T            (left camera)
#define VT_A0R 24     // ric^T Rj^T Ri
#define VT_A0RC 34    // ric^T Rj^T Ri ric
#define VT_A1 44      // ric2^T Rj^T           (right camera)
#define VT_A1R 54     // ric2^T Rj^T Ri
#define VT_A1RC 64    // ric2^T Rj^T Ri ric
#define VT_N 74       // 592 B: the four segments of a wave sit 20 banks apart (no conflict between their 16-byte reads)

// a0 b0 + a1 b1 with the rounding spelled out (one product, then one fused multiply-add): the landmark-side terms are formed by three
// kernels (single wave, frame-parallel, producer / consumer) that must agree bit for bit, and left to the optimiser each of them may
// contract the expression the other way round
VD double dot2(double a0, double b0, double a1, double b1) { return fma(a0, b0, a1 * b1); }

struct Red3 { double r00, r02, r12; };   // reduce = [r00 0 r02; 0 r00 r12] (already times sqrt_info and sqrt(rho'))

// T = reduce * M, M 3 x 3 row-major
VD void red_mul(const Red3 &R, const double *M, double *t0, double *t1) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    t0[c] = R.r00 * M[c] + R.r02 * M[6 + c];
    t1[c] = R.r00 * M[3 + c] + R.r12 * M[6 + c];
  }
}
// T = reduce * M^T
VD void red_mul_t(const Red3 &R, const double *M, double *t0, double *t1) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    t0[c] = R.r00 * M[3 * c] + R.r02 * M[3 * c + 2];
    t1[c] = R.r00 * M[3 * c + 1] + R.r12 * M[3 * c + 2];
  }
}
// out = a x b
VD void cross3(const double *a, const v3 &b, double *o) {
  o[0] = a[1] * b.z - a[2] * b.y;
  o[1] = a[2] * b.x - a[0] * b.z;
  o[2] = a[0] * b.y - a[1] * b.x;
}
VD void cross3(const v3 &a, const double *b, double *o) {
  o[0] = a.y * b[2] - a.z * b[1];
  o[1] = a.z * b[0] - a.x * b[2];
  o[2] = a.x * b[1] - a.y * b[0];
}
// rows of reduce * skew(v)
VD void red_skew(const Red3 &R, const v3 &v, double *o0, double *o1) {
  o0[0] = -R.r02 * v.y; o0[1] = R.r02 * v.x - R.r00 * v.z; o0[2] = R.r00 * v.y;
  o1[0] = R.r00 * v.z - R.r12 * v.y; o1[1] = R.r12 * v.x; o1[2] = -R.r00 * v.x;
}

// What a landmark keeps across the frames that observe it.
struct VisLane {
  v3 pci;       // pts_camera_i = pts_i_td / lambda
  v3 p_i;       // pts_imu_i
  v3 p_w;       // Ri pts_imu_i + Pi
  double inv_lam, vix, viy;
};

// residual of a projected point + the Huber weight: returns rho(s); R = sqrt(rho') sq * d(pi)/d(pts_camera_j), r scaled by sqrt(rho')
VD double vis_residual(const v3 &pcj, double ptx, double pty, double sq, double huber_a, double *r, Red3 &R, double &sqw) {
  const double inv_z = 1.0 / pcj.z;
  const double r0 = sq * (pcj.x * inv_z - ptx), r1 = sq * (pcj.y * inv_z - pty);
  const double s = r0 * r0 + r1 * r1, b = huber_a * huber_a;
  double rho0 = s, w = 1.0;
  if (s > b) {
    const double rt = sqrt(s);
    rho0 = 2.0 * huber_a * rt - b;
    double rho1 = huber_a / rt;
    if (rho1 < 2.2250738585072014e-308) rho1 = 2.2250738585072014e-308;
    w = sqrt(rho1);
  }
  sqw = sq * w;
  R.r00 = sqw * inv_z;
  R.r02 = -R.r00 * pcj.x * inv_z;
  R.r12 = -R.r00 * pcj.y * inv_z;
  r[0] = r0 * w; r[1] = r1 * w;
  return rho0;
}

// ProjectionTwoFrameOneCamFactor (CAM 0) / ProjectionTwoFrameTwoCamFactor (CAM 1) of one landmark seen from frame j.
// x0 / x1: the two corrected rows in the Gram slot's column order (vilo_internal.hpp: GC_T 3 | GC_RI 3 | GC_RJ 3 | GC_E0 6 | GC_R |
// GC_E1 6 | GC_TD; d r / d P_j = -d r / d P_i is not stored); Jl: d r / d lambda. Returns rho(s).
// ob: pts_j (3), vel_j (2) of this camera; dtj = td - td_j.
template <int CAM>
VD double vis_two_frame(const double *wt, const double *tb, const VisLane &L, const v3 &p_j, const double *ob, double dtj, double sq,
                        double huber_a, double *x0, double *x1, double *Jl) {
  const double *ricK = wt + (CAM ? VW_RIC2 : VW_RIC), *ticK = wt + (CAM ? VW_TIC2 : VW_TIC);
  const double *A = tb + (CAM ? VT_A1 : VT_A0), *AR = tb + (CAM ? VT_A1R : VT_A0R), *ARC = tb + (CAM ? VT_A1RC : VT_A0RC);
  // pts_camera_j = ricK^T (pts_imu_j - ticK)
  const v3 d = mk3(p_j.x - ticK[0], p_j.y - ticK[1], p_j.z - ticK[2]);
  const v3 pcj = mk3(ricK[0] * d.x + ricK[3] * d.y + ricK[6] * d.z, ricK[1] * d.x + ricK[4] * d.y + ricK[7] * d.z,
                     ricK[2] * d.x + ricK[5] * d.y + ricK[8] * d.z);
  Red3 R;
  double sqw, r[2];
  const double rho0 = vis_residual(pcj, ob[0] - ob[3] * dtj, ob[1] - ob[4] * dtj, sq, huber_a, r, R, sqw);
  double t0[3], t1[3];
  // pose_i: [A | -A Ri skew(pts_imu_i)] (pose_j translation: -A)
  red_mul(R, A, x0 + GC_T, x1 + GC_T);
  double e0[3], e1[3];
  red_mul(R, AR, e0, e1);
  cross3(L.p_i, e0, x0 + GC_RI);
  cross3(L.p_i, e1, x1 + GC_RI);
  // pose_j rotation: ricK^T skew(pts_imu_j)
  red_mul_t(R, ricK, t0, t1);
  cross3(t0, p_j, x0 + GC_RJ);
  cross3(t1, p_j, x1 + GC_RJ);
  double c0[3], c1[3];
  red_mul(R, ARC, c0, c1);
  if (CAM == 0) {
    // ex0: [ric^T (Rj^T Ri - I) | -tmp_r skew(pts_camera_i) + skew(pts_camera_j)], no ex1 block
    double s0[3], s1[3];
    red_skew(R, pcj, s0, s1);
    cross3(L.pci, c0, x0 + GC_E0 + 3);
    cross3(L.pci, c1, x1 + GC_E0 + 3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      x0[GC_E0 + c] = e0[c] - t0[c]; x1[GC_E0 + c] = e1[c] - t1[c];
      x0[GC_E0 + 3 + c] += s0[c]; x1[GC_E0 + 3 + c] += s1[c];
      x0[GC_E1 + c] = 0.0; x1[GC_E1 + c] = 0.0; x0[GC_E1 + 3 + c] = 0.0; x1[GC_E1 + 3 + c] = 0.0;
    }
  } else {
    // ex0: [A Ri | -A Ri ric skew(pts_camera_i)], ex1: [-ric2^T | skew(pts_camera_j)]
    cross3(L.pci, c0, x0 + GC_E0 + 3);
    cross3(L.pci, c1, x1 + GC_E0 + 3);
    red_skew(R, pcj, x0 + GC_E1 + 3, x1 + GC_E1 + 3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      x0[GC_E0 + c] = e0[c]; x1[GC_E0 + c] = e1[c];
      x0[GC_E1 + c] = -t0[c]; x1[GC_E1 + c] = -t1[c];
    }
  }
  // lambda: reduce * (tmp_r pts_i_td) * (-1 / lambda^2);  td: reduce * (tmp_r vel_i) * (-1 / lambda) + sqrt_info vel_j
  const double nil = -L.inv_lam;
  Jl[0] = nil * (c0[0] * L.pci.x + c0[1] * L.pci.y + c0[2] * L.pci.z);
  Jl[1] = nil * (c1[0] * L.pci.x + c1[1] * L.pci.y + c1[2] * L.pci.z);
  x0[GC_TD] = nil * (c0[0] * L.vix + c0[1] * L.viy) + sqw * ob[3];
  x1[GC_TD] = nil * (c1[0] * L.vix + c1[1] * L.viy) + sqw * ob[4];
  x0[GC_R] = r[0]; x1[GC_R] = r[1];
  return rho0;
}

// ProjectionOneFrameTwoCamFactor: the right-camera observation in the start frame (ex0, ex1, lambda, td only; the pose columns are zero).
// pts_i: the un-shifted left observation (the factor's lambda Jacobian uses pts_i, not pts_i_td: projectionOneFrameTwoCamFactor.cpp:119).
VD double vis_one_frame(const double *wt, const VisLane &L, const v3 &pts_i, const double *ob, double dtj, double sq, double huber_a, double *x0,
                        double *x1, double *Jl) {
  const double *ric2 = wt + VW_RIC2, *tic2 = wt + VW_TIC2, *A = wt + VW_A2;
  const v3 d = mk3(L.p_i.x - tic2[0], L.p_i.y - tic2[1], L.p_i.z - tic2[2]);
  const v3 pcj = mk3(ric2[0] * d.x + ric2[3] * d.y + ric2[6] * d.z, ric2[1] * d.x + ric2[4] * d.y + ric2[7] * d.z,
                     ric2[2] * d.x + ric2[5] * d.y + ric2[8] * d.z);
  Red3 R;
  double sqw, r[2];
  const double rho0 = vis_residual(pcj, ob[0] - ob[3] * dtj, ob[1] - ob[4] * dtj, sq, huber_a, r, R, sqw);
  double t0[3], t1[3], c0[3], c1[3];
  red_mul_t(R, ric2, t0, t1);   // reduce * ric2^T
  red_mul(R, A, c0, c1);        // reduce * ric2^T ric
#pragma unroll
  for (int c = 0; c < GC_E0; ++c) { x0[c] = 0.0; x1[c] = 0.0; }
  cross3(L.pci, c0, x0 + GC_E0 + 3);
  cross3(L.pci, c1, x1 + GC_E0 + 3);
  red_skew(R, pcj, x0 + GC_E1 + 3, x1 + GC_E1 + 3);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    x0[GC_E0 + c] = t0[c]; x1[GC_E0 + c] = t1[c];
    x0[GC_E1 + c] = -t0[c]; x1[GC_E1 + c] = -t1[c];
  }
  const double il2 = -(L.inv_lam * L.inv_lam), nil = -L.inv_lam;
  Jl[0] = il2 * (c0[0] * pts_i.x + c0[1] * pts_i.y + c0[2] * pts_i.z);
  Jl[1] = il2 * (c1[0] * pts_i.x + c1[1] * pts_i.y + c1[2] * pts_i.z);
  x0[GC_TD] = nil * (c0[0] * L.vix + c0[1] * L.viy) + sqw * ob[3];
  x1[GC_TD] = nil * (c1[0] * L.vix + c1[1] * L.viy) + sqw * ob[4];
  x0[GC_R] = r[0]; x1[GC_R] = r[1];
  return rho0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Compact 16-column form of the same rows (solve passes while td is a constant block: estimate_td 0, estimator.cpp:1104).
// The translation columns of BOTH extrinsics are the pose-translation columns times a rotation that depends on the (start frame,
// observing frame) pair only — with B = d r / d P_i = reduce ricK^T Rj^T:
//     TwoFrameOneCam:  d r / d tic  = reduce ric^T (Rj^T Ri - I) = B (Ri - Rj)      (projectionTwoFrameOneCamFactor.cpp:109-118)
//     TwoFrameTwoCam:  d r / d tic  = reduce ric2^T Rj^T Ri      = B Ri,   d r / d tic2 = -reduce ric2^T = -B Rj   (…TwoCamFactor.cpp:114-132)
//     OneFrameTwoCam:  d r / d tic  = reduce ric2^T =: B,        d r / d tic2 = -B   (…OneFrameTwoCamFactor.cpp:92-108; no pose columns)
// so their Gram blocks are 3 x 3 transforms of the B blocks, applied once per (chunk, frame) slot by k_assemble instead of once per
// factor here: a row has 16 columns = ONE FP64-MFMA tile per camera (the 23-column form needs three for a right-camera factor).
//   GK_B 3 | GK_RI 3 (theta_i) | GK_RJ 3 (theta_j) | GK_C0 3 (theta_ic) | GK_C1 3 (theta_ic2; zero for the left camera) | GK_R
// A slot stores the upper triangle of G = C0 + C1 (the Grams of the left / right camera rows; 136) and rows 0..2 (B) of C1 (48):
// C0's B rows are G's minus C1's.
#define GK_B 0
#define GK_RI 3
#define GK_RJ 6
#define GK_C0 9
#define GK_C1 12
#define GK_R 15
#define VILO_GKC 16
#define VILO_GRAMC_TRI 136
#define VILO_GRAMC 184
VD int tri16(int a, int b) { return a * 16 - (a * (a - 1)) / 2 + (b - a); }   // a <= b

// tc: the extrinsic translation columns of the two rows (only the landmark's coupling row needs them per factor): tic (row 0, row 1), tic2 (row 0, row 1)
template <int CAM>
VD double vis_two_frame_c(const double *wt, const double *tb, const VisLane &L, const v3 &p_j, const double *ob, double dtj, double sq,
                          double huber_a, double *x0, double *x1, double *Jl, double tc[4][3]) {
  const double *ricK = wt + (CAM ? VW_RIC2 : VW_RIC), *ticK = wt + (CAM ? VW_TIC2 : VW_TIC);
  const double *A = tb + (CAM ? VT_A1 : VT_A0), *AR = tb + (CAM ? VT_A1R : VT_A0R), *ARC = tb + (CAM ? VT_A1RC : VT_A0RC);
  const v3 d = mk3(p_j.x - ticK[0], p_j.y - ticK[1], p_j.z - ticK[2]);
  const v3 pcj = mk3(ricK[0] * d.x + ricK[3] * d.y + ricK[6] * d.z, ricK[1] * d.x + ricK[4] * d.y + ricK[7] * d.z,
                     ricK[2] * d.x + ricK[5] * d.y + ricK[8] * d.z);
  Red3 R;
  double sqw, r[2];
  const double rho0 = vis_residual(pcj, ob[0] - ob[3] * dtj, ob[1] - ob[4] * dtj, sq, huber_a, r, R, sqw);
  double t0[3], t1[3], e0[3], e1[3], c0[3], c1[3];
  red_mul(R, A, x0 + GK_B, x1 + GK_B);
  red_mul(R, AR, e0, e1);
  cross3(L.p_i, e0, x0 + GK_RI);
  cross3(L.p_i, e1, x1 + GK_RI);
  red_mul_t(R, ricK, t0, t1);
  cross3(t0, p_j, x0 + GK_RJ);
  cross3(t1, p_j, x1 + GK_RJ);
  red_mul(R, ARC, c0, c1);
  cross3(L.pci, c0, x0 + GK_C0);
  cross3(L.pci, c1, x1 + GK_C0);
  if (CAM == 0) {
    double s0[3], s1[3];
    red_skew(R, pcj, s0, s1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      x0[GK_C0 + c] += s0[c]; x1[GK_C0 + c] += s1[c];
      x0[GK_C1 + c] = 0.0; x1[GK_C1 + c] = 0.0;
      tc[0][c] = e0[c] - t0[c]; tc[1][c] = e1[c] - t1[c];
      tc[2][c] = 0.0; tc[3][c] = 0.0;
    }
  } else {
    red_skew(R, pcj, x0 + GK_C1, x1 + GK_C1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      tc[0][c] = e0[c]; tc[1][c] = e1[c];
      tc[2][c] = -t0[c]; tc[3][c] = -t1[c];
    }
  }
  const double nil = -L.inv_lam;
  Jl[0] = nil * (c0[0] * L.pci.x + c0[1] * L.pci.y + c0[2] * L.pci.z);
  Jl[1] = nil * (c1[0] * L.pci.x + c1[1] * L.pci.y + c1[2] * L.pci.z);
  x0[GK_R] = r[0]; x1[GK_R] = r[1];
  return rho0;
}

// OneFrameTwoCam in the compact form: B = reduce ric2^T (tic: +B, tic2: -B; k_assemble uses identity transforms for a frame-0 slot)
VD double vis_one_frame_c(const double *wt, const VisLane &L, const v3 &pts_i, const double *ob, double dtj, double sq, double huber_a, double *x0,
                          double *x1, double *Jl, double tc[4][3]) {
  const double *ric2 = wt + VW_RIC2, *tic2 = wt + VW_TIC2, *A = wt + VW_A2;
  const v3 d = mk3(L.p_i.x - tic2[0], L.p_i.y - tic2[1], L.p_i.z - tic2[2]);
  const v3 pcj = mk3(ric2[0] * d.x + ric2[3] * d.y + ric2[6] * d.z, ric2[1] * d.x + ric2[4] * d.y + ric2[7] * d.z,
                     ric2[2] * d.x + ric2[5] * d.y + ric2[8] * d.z);
  Red3 R;
  double sqw, r[2];
  const double rho0 = vis_residual(pcj, ob[0] - ob[3] * dtj, ob[1] - ob[4] * dtj, sq, huber_a, r, R, sqw);
  double c0[3], c1[3];
  red_mul_t(R, ric2, x0 + GK_B, x1 + GK_B);   // reduce * ric2^T
  red_mul(R, A, c0, c1);                      // reduce * ric2^T ric
  cross3(L.pci, c0, x0 + GK_C0);
  cross3(L.pci, c1, x1 + GK_C0);
  red_skew(R, pcj, x0 + GK_C1, x1 + GK_C1);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    x0[GK_RI + c] = 0.0; x1[GK_RI + c] = 0.0; x0[GK_RJ + c] = 0.0; x1[GK_RJ + c] = 0.0;
    tc[0][c] = x0[GK_B + c]; tc[1][c] = x1[GK_B + c];
    tc[2][c] = -x0[GK_B + c]; tc[3][c] = -x1[GK_B + c];
  }
  const double il2 = -(L.inv_lam * L.inv_lam);
  Jl[0] = il2 * (c0[0] * pts_i.x + c0[1] * pts_i.y + c0[2] * pts_i.z);
  Jl[1] = il2 * (c1[0] * pts_i.x + c1[1] * pts_i.y + c1[2] * pts_i.z);
  x0[GK_R] = r[0]; x1[GK_R] = r[1];
  return rho0;
}

// One row (r) of the products of one camera (kind) for the pair (start frame s, observing frame j); kind 0 also stores Rj and Pj.
// xs: the window's state in vector2double order; wt: the window-level table.
VD void vis_build_pair_row(const double *xs, const double *wt, int s, int j, int kind, int r, double *tb) {
  const double *pi_ = xs + 7 * s, *pj_ = xs + 7 * j;   // XO_POSE = 0
  const m3 Ri = qR(ldq_pose(pi_)), Rj = qR(ldq_pose(pj_));
  const double *ricK = wt + (kind ? VW_RIC2 : VW_RIC), *ric = wt + VW_RIC;
  const double k0 = ricK[r], k1 = ricK[3 + r], k2 = ricK[6 + r];   // column r of ricK = row r of ricK^T
  double a[3], ar[3], arc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) a[c] = k0 * Rj.a[3 * c] + k1 * Rj.a[3 * c + 1] + k2 * Rj.a[3 * c + 2];        // (ricK^T Rj^T)[r][c]
#pragma unroll
  for (int c = 0; c < 3; ++c) ar[c] = a[0] * Ri.a[c] + a[1] * Ri.a[3 + c] + a[2] * Ri.a[6 + c];
#pragma unroll
  for (int c = 0; c < 3; ++c) arc[c] = ar[0] * ric[c] + ar[1] * ric[3 + c] + ar[2] * ric[6 + c];
  double *o = tb + (kind ? VT_A1 : VT_A0) + 3 * r;
#pragma unroll
  for (int c = 0; c < 3; ++c) { o[c] = a[c]; o[10 + c] = ar[c]; o[20 + c] = arc[c]; }
  if (kind == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) tb[VT_RJ + 3 * r + c] = (r == 0) ? Rj.a[c] : (r == 1 ? Rj.a[3 + c] : Rj.a[6 + c]);
    tb[VT_PJ + r] = pj_[r];
  }
}

}  // namespace vilo
