// TEST INFRASTRUCTURE: compiles the product's inline device math (cerberus_amd/csrc/factors.hpp,
// vilo_math.hpp) for the HOST so the `-m "not gpu"` suite can compare it with the oracle before any GPU
// minute is spent. This library is never loaded by the product (libvilo_gpu.so has no CPU path).
#include "../../cerberus_amd/csrc/factors.hpp"
#include "../../cerberus_amd/csrc/visual_lin.hpp"
#include "../../cerberus_amd/csrc/assemble_compact.hpp"
#include "../../include/vilo_gpu.h"

using namespace vilo;

extern "C" {
// local-parameterisation Jacobians: J_* are 2x6 / 2 doubles
void hc_proj(int kind, const double *obs12, const double *pose_i, const double *pose_j, const double *ex0, const double *ex1,
             double inv_dep, double td, double sq, double *r, int want_jac, double *J_i, double *J_j, double *J_e0, double *J_e1,
             double *J_l, double *J_td) {
  if (kind == 0) proj_factor<0>(obs12, pose_i, pose_j, ex0, ex1, inv_dep, td, sq, r, want_jac, J_i, J_j, J_e0, J_e1, J_l, J_td);
  else if (kind == 1) proj_factor<1>(obs12, pose_i, pose_j, ex0, ex1, inv_dep, td, sq, r, want_jac, J_i, J_j, J_e0, J_e1, J_l, J_td);
  else proj_factor<2>(obs12, pose_i, pose_j, ex0, ex1, inv_dep, td, sq, r, want_jac, J_i, J_j, J_e0, J_e1, J_l, J_td);
}
// corrected (r, one Jacobian column) for a 2-row block; returns rho0
double hc_correct(double a, double *r2, double *jcol2) {
  const double s = r2[0] * r2[0] + r2[1] * r2[1];
  Corrector c = make_corrector(a, s);
  correct_col(c, r2[0], r2[1], jcol2[0], jcol2[1]);
  r2[0] *= c.residual_scaling;
  r2[1] *= c.residual_scaling;
  return c.rho0;
}
void hc_imu_leg_raw(const vilo_preint *pre, double g_norm, const double *pose_i, const double *sb_i, const double *lb_i,
                    const double *pose_j, const double *sb_j, const double *lb_j, double *r31, double *J31x38) {
  PreintHead h;
  fill_preint_head(*pre, h);
  for (int i = 0; i < 31 * 38; ++i) J31x38[i] = 0.0;
  imu_leg_raw(h, g_norm, pose_i, sb_i, lb_i, pose_j, sb_j, lb_j, r31, true, J31x38, 38);
}
void hc_imu_raw(const vilo_preint_imu *pre, double g_norm, const double *pose_i, const double *sb_i, const double *pose_j,
                const double *sb_j, double *r15, double *J15x30) {
  PreintHead h;
  fill_preint_head_imu(*pre, h);
  for (int i = 0; i < 15 * 30; ++i) J15x30[i] = 0.0;
  imu_raw(h, g_norm, pose_i, sb_i, pose_j, sb_j, r15, true, J15x30, 30);
}
void hc_leg_kin(const double *q, double lc, const double *rf, double *f3, double *J9, double *dfdrho3, double *dJ27, double *dJdrho9) {
  LegKin k;
  leg_kin_full(q, lc, rf, k);
  st3(f3, k.f);
  st3(dfdrho3, k.df_drho);
  for (int i = 0; i < 9; ++i) { J9[i] = k.J.a[i]; dJdrho9[i] = k.dJ_drho.a[i]; }
  for (int n = 0; n < 3; ++n)
    for (int i = 0; i < 9; ++i) dJ27[9 * n + i] = k.dJ[n].a[i];
}
void hc_pose_plus(const double *x, const double *d, double *out) { pose_plus(x, d, out); }
void hc_prior_dx(const double *x, const double *x0, int size, double *dx) { prior_dx(x, x0, size, dx); }

// The fused linearisation's factor forms (visual_lin.hpp: hoisted rotation products, Huber weight folded into `reduce`) for one landmark:
// kind 0 / 1 / 2 as in hc_proj; x0 / x1: the two corrected rows in the Gram slot's 23-column order (factors.hpp: GC_T 3 | GC_RI 3 | GC_RJ 3 |
// GC_E0 6 | GC_R | GC_E1 6 | GC_TD; d r / d P_j = -d r / d P_i is not stored); returns rho(s).
// The tables are built the way k_visual_linearize builds them (window table, then the 2 x 3 rows of the pair's table).
double hc_vis_lin(int kind, const double *obs12, const double *pose_i, const double *pose_j, const double *ex0, const double *ex1,
                  double inv_dep, double td, double sq, double huber_a, double *x0, double *x1, double *Jl) {
  double xs[14], wt[VW_N], tb[VT_N];
  for (int i = 0; i < 7; ++i) { xs[i] = pose_i[i]; xs[7 + i] = pose_j[i]; }
  const m3 ric = qR(ldq_pose(ex0)), ric2 = qR(ldq_pose(ex1));
  const m3 A2 = tr(ric2) * ric;
  for (int q = 0; q < 9; ++q) { wt[VW_RIC + q] = ric.a[q]; wt[VW_RIC2 + q] = ric2.a[q]; wt[VW_A2 + q] = A2.a[q]; }
  for (int q = 0; q < 3; ++q) { wt[VW_TIC + q] = ex0[q]; wt[VW_TIC2 + q] = ex1[q]; }
  for (int k = 0; k < 2; ++k)
    for (int r = 0; r < 3; ++r) vis_build_pair_row(xs, wt, 0, 1, k, r, tb);
  VisLane L;
  const double dti = td - obs12[10];
  L.inv_lam = 1.0 / inv_dep;
  L.vix = obs12[6]; L.viy = obs12[7];
  L.pci = mk3((obs12[0] - obs12[6] * dti) * L.inv_lam, (obs12[1] - obs12[7] * dti) * L.inv_lam, obs12[2] * L.inv_lam);
  L.p_i = ric * L.pci + ld3(ex0);
  L.p_w = qrot(ldq_pose(pose_i), L.p_i) + ld3(pose_i);
  const v3 d = mk3(L.p_w.x - tb[VT_PJ], L.p_w.y - tb[VT_PJ + 1], L.p_w.z - tb[VT_PJ + 2]);
  const v3 p_j = mk3(tb[0] * d.x + tb[3] * d.y + tb[6] * d.z, tb[1] * d.x + tb[4] * d.y + tb[7] * d.z, tb[2] * d.x + tb[5] * d.y + tb[8] * d.z);
  const double obc[5] = {obs12[3], obs12[4], obs12[5], obs12[8], obs12[9]};
  const double dtj = td - obs12[11];
  if (kind == 0) return vis_two_frame<0>(wt, tb, L, p_j, obc, dtj, sq, huber_a, x0, x1, Jl);
  if (kind == 1) return vis_two_frame<1>(wt, tb, L, p_j, obc, dtj, sq, huber_a, x0, x1, Jl);
  return vis_one_frame(wt, L, mk3(obs12[0], obs12[1], obs12[2]), obc, dtj, sq, huber_a, x0, x1, Jl);
}

// The compact 16-column form of the same rows (visual_lin.hpp: GK_B 3 | GK_RI 3 | GK_RJ 3 | GK_C0 3 | GK_C1 3 | GK_R) and the extrinsic
// translation columns tc[4][3] (tic row 0 / 1, tic2 row 0 / 1) the landmark's coupling row needs.
double hc_vis_lin_c(int kind, const double *obs12, const double *pose_i, const double *pose_j, const double *ex0, const double *ex1,
                    double inv_dep, double td, double sq, double huber_a, double *x0, double *x1, double *Jl, double *tc12) {
  double xs[14], wt[VW_N], tb[VT_N];
  for (int i = 0; i < 7; ++i) { xs[i] = pose_i[i]; xs[7 + i] = pose_j[i]; }
  const m3 ric = qR(ldq_pose(ex0)), ric2 = qR(ldq_pose(ex1));
  const m3 A2 = tr(ric2) * ric;
  for (int q = 0; q < 9; ++q) { wt[VW_RIC + q] = ric.a[q]; wt[VW_RIC2 + q] = ric2.a[q]; wt[VW_A2 + q] = A2.a[q]; }
  for (int q = 0; q < 3; ++q) { wt[VW_TIC + q] = ex0[q]; wt[VW_TIC2 + q] = ex1[q]; }
  for (int k = 0; k < 2; ++k)
    for (int r = 0; r < 3; ++r) vis_build_pair_row(xs, wt, 0, 1, k, r, tb);
  VisLane L;
  const double dti = td - obs12[10];
  L.inv_lam = 1.0 / inv_dep;
  L.vix = obs12[6]; L.viy = obs12[7];
  L.pci = mk3((obs12[0] - obs12[6] * dti) * L.inv_lam, (obs12[1] - obs12[7] * dti) * L.inv_lam, obs12[2] * L.inv_lam);
  L.p_i = ric * L.pci + ld3(ex0);
  L.p_w = qrot(ldq_pose(pose_i), L.p_i) + ld3(pose_i);
  const v3 d = mk3(L.p_w.x - tb[VT_PJ], L.p_w.y - tb[VT_PJ + 1], L.p_w.z - tb[VT_PJ + 2]);
  const v3 p_j = mk3(tb[0] * d.x + tb[3] * d.y + tb[6] * d.z, tb[1] * d.x + tb[4] * d.y + tb[7] * d.z, tb[2] * d.x + tb[5] * d.y + tb[8] * d.z);
  const double obc[5] = {obs12[3], obs12[4], obs12[5], obs12[8], obs12[9]};
  const double dtj = td - obs12[11];
  double tc[4][3];
  double rho;
  if (kind == 0) rho = vis_two_frame_c<0>(wt, tb, L, p_j, obc, dtj, sq, huber_a, x0, x1, Jl, tc);
  else if (kind == 1) rho = vis_two_frame_c<1>(wt, tb, L, p_j, obc, dtj, sq, huber_a, x0, x1, Jl, tc);
  else rho = vis_one_frame_c(wt, L, mk3(obs12[0], obs12[1], obs12[2]), obc, dtj, sq, huber_a, x0, x1, Jl, tc);
  for (int i = 0; i < 12; ++i) tc12[i] = tc[i / 3][i % 3];
  return rho;
}

// k_assemble's compact visual part, its 256 threads emulated one after the other: H (80 x 80 row-major, lower triangle + mirrored
// diagonal blocks are NOT formed: entry (hi, lo) only) and the gradient g (80). poses: [11][7] of the window's frames.
void hc_assemble_compact(int nch, const unsigned *chunk_tab, const double *slots, const double *poses, double *H80, double *g80) {
  double Rt[12 * 9];
  for (int f = 0; f < 11; ++f) {
    const m3 R = qR(ldq_pose(poses + 7 * f));
    for (int q = 0; q < 9; ++q) Rt[9 * f + q] = R.a[q];
  }
  for (int q = 0; q < 9; ++q) Rt[99 + q] = (q % 4 == 0) ? 1.0 : 0.0;
  for (int ch = 0; ch < nch; ++ch) {
    const int s = chunk_tab[ch] & 255, km = (chunk_tab[ch] >> 8) & 255, sl0 = chunk_tab[ch] >> 16;
    auto rmw = [&](int hi, int lo, double v) { H80[hi * 80 + lo] += v; };
    for (int tid = 0; tid < 256; ++tid)
      assemble_visual_compact_chunk(tid, s, km, slots + (size_t)sl0 * VILO_GRAMC, Rt, rmw, [&](int cd, double v) { g80[cd] += v; });
    for (int q = 0; q < 21; ++q) {   // wave 1 of the kernel: three lanes per entry, partial sums added in lane order
      double p[3];
      for (int grp = 0; grp < 3; ++grp) p[grp] = ac_t8_partial(q, grp, 3, s, km, slots + (size_t)sl0 * VILO_GRAMC, Rt);
      ac_t8_apply(q, (p[0] + p[1]) + p[2], rmw);
    }
  }
}
}
