// TEST INFRASTRUCTURE: compiles the product's inline device math (cerberus_amd/csrc/factors.hpp,
// vilo_math.hpp) for the HOST so the `-m "not gpu"` suite can compare it with the oracle before any GPU
// minute is spent. This library is never loaded by the product (libvilo_gpu.so has no CPU path).
#include "../../cerberus_amd/csrc/factors.hpp"
#include "../../cerberus_amd/csrc/visual_lin.hpp"
#include "../../cerberus_amd/csrc/assemble_compact.hpp"
#include "../../cerberus_amd/csrc/preint_blocks.hpp"
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
// imu_fused_body's way to the same [J | r] (kernels_solve.hip): the seven 3 x 3 matrices and the residual from imu_blocks, every entry of the
// 32 x 48 operand image fetched through imu_gather_table — emulated entry by entry. leg: 1 IMULegFactor, 0 IMUFactor embedded in the
// 38-column layout (frame-j blocks from column 19). out: 32 x 48 row-major.
void hc_imu_blocks_gather(const vilo_preint *pre, const vilo_preint_imu *pre_imu, int leg, double g_norm, const double *pose_i, const double *sb_i,
                          const double *lb_i, const double *pose_j, const double *sb_j, const double *lb_j, double *out32x48) {
  PreintHead h;
  if (leg) fill_preint_head(*pre, h); else fill_preint_head_imu(*pre_imu, h);
  double pool[IB_N];
  for (int i = 0; i < IB_N; ++i) pool[i] = 0.0;
  const double T = imu_blocks(h, g_norm, leg != 0, pose_i, sb_i, lb_i, pose_j, sb_j, lb_j, pool);
  static const ImuGatherTable tabs[2] = {imu_gather_table(false), imu_gather_table(true)};
  const double *hd = (const double *)&h;
  for (int e = 0; e < 32 * 48; ++e) {
    const unsigned g = tabs[leg ? 1 : 0].e[e], code = g >> 12;
    const double val = ((g & 0x100) ? hd : pool)[g & 0xff];
    const double cf = code == 1 ? 1.0 : (code == 2 ? -1.0 : (code == 3 ? T : -T));
    out32x48[e] = code ? cf * val : 0.0;
  }
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
    for (int q = 0; q < 18; ++q) {   // wave 2 of the kernel: two lanes per entry (odd / even frames)
      double s5[2], s6[2];
      for (int par = 0; par < 2; ++par) ac_t56_partial(q, par, s, km, slots + (size_t)sl0 * VILO_GRAMC, Rt, rmw, s5[par], s6[par]);
      ac_t56_apply(q, s, s5[0] + s5[1], s6[0] + s6[1], rmw);
    }
  }
}

// The same through the PASS form of the full batch's pose assembly (kernels_asm_full.hip: passes of <= AC_PASS frames through a six-slot
// stage; assemble_compact.hpp's ac_pass_* bodies): the pass split, the four waves' bodies and the lane groups of T8 / T5 + T6 emulated as the
// kernel runs them.
void hc_assemble_compact_passes(int nch, const unsigned *chunk_tab, const double *slots, const double *poses, double *H80, double *g80) {
  double Rt[12 * 9];
  for (int f = 0; f < 11; ++f) {
    const m3 R = qR(ldq_pose(poses + 7 * f));
    for (int q = 0; q < 9; ++q) Rt[9 * f + q] = R.a[q];
  }
  for (int q = 0; q < 9; ++q) Rt[99 + q] = (q % 4 == 0) ? 1.0 : 0.0;
  auto rmw = [&](int hi, int lo, double v) { H80[hi * 80 + lo] += v; };
  auto gadd = [&](int cd, double v) { g80[cd] += v; };
  for (int ch = 0; ch < nch; ++ch) {
    const int s = chunk_tab[ch] & 255, km = (chunk_tab[ch] >> 8) & 255, sl0 = chunk_tab[ch] >> 16;
    const int npass = (km + AC_PASS - 1) / AC_PASS, base = km / npass, rem = km - base * npass;
    for (int i = 0; i < npass; ++i) {
      const int t0 = i * base + (i < rem ? i : rem), np = base + (i < rem ? 1 : 0);
      double ps[(AC_PASS + 1) * VILO_GRAMC];   // the kernel's stage: the pass's slots, garbage up to AC_PASS, then the slot of zeros
      for (int e = 0; e < (AC_PASS + 1) * VILO_GRAMC; ++e) ps[e] = (e < np * VILO_GRAMC) ? slots[(size_t)(sl0 + t0) * VILO_GRAMC + e] : (e < AC_PASS * VILO_GRAMC ? 1e300 : 0.0);
      for (int tid = 0; tid < 256; ++tid) assemble_visual_compact_pass(tid, s, t0, np, ps, Rt, rmw, gadd);
      for (int q = 0; q < 21; ++q) {
        double p[3];
        for (int grp = 0; grp < 3; ++grp) p[grp] = ac_t8_pass(q, grp, s, t0, np, ps, Rt);
        ac_t8_apply(q, (p[0] + p[1]) + p[2], rmw);
      }
      for (int q = 0; q < 18; ++q) {
        double s5[2], s6[2];
        for (int par = 0; par < 2; ++par) ac_t56_pass(q, par, s, t0, np, ps, Rt, rmw, s5[par], s6[par]);
        ac_t56_apply(q, s, s5[0] + s5[1], s6[0] + s6[1], rmw);
      }
    }
  }
}

// dF = F - I (32 x 31, row-major) and V (32 x 48) of one midpoint step of IMULegIntegrationBase.
// in: R0[9] R1[9] un_gyr[3] a0[3] a1[3] Rbr[9] dt, then per (leg j, endpoint e), q = 2 j + e: v[3] p[3] h0[9] J[9] g0[3]  (27 doubles each).
// mode 0: the blocks written out one after the other as the reference writes them (imu_leg_integration_base.cpp:376-465), 3 x 3
// temporaries; mode 1: the product's lane-parallel construction (preint_blocks.hpp), its 64 lanes emulated round by round.
void hc_preint_blocks(int mode, const double *in, double *dF, double *V) {
  const m3 R0 = ld_m3_rowmajor(in), R1 = ld_m3_rowmajor(in + 9), Rbr = ld_m3_rowmajor(in + 27);
  const v3 un_gyr = ld3(in + 18), a0 = ld3(in + 21), a1 = ld3(in + 24);
  const double dt = in[36];
  const double *legs = in + 37;
  const m3 I3 = m3_eye();
  const m3 Rwx = skew(un_gyr), Ra0 = skew(a0), Ra1 = skew(a1);
  const m3 kappa_7 = I3 - Rwx * dt;
  for (int i = 0; i < 32 * 31; ++i) dF[i] = 0.0;
  for (int i = 0; i < 32 * 48; ++i) V[i] = 0.0;
  if (mode == 0) {
    auto putF = [&](int r0, int c0, const m3 &A) { for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) dF[(r0 + a) * 31 + c0 + b] = A.a[3 * a + b]; };
    auto putV = [&](int r0, int c0, const m3 &A) { for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) V[(r0 + a) * 48 + c0 + b] = A.a[3 * a + b]; };
    const m3 kappa_1 = (R0 * Ra0) * (-0.5 * dt) + (R1 * Ra1 * kappa_7) * (-0.5 * dt);
    putF(0, 3, kappa_1 * (0.5 * dt));
    putF(0, 6, I3 * dt);
    putF(0, 21, (R0 + R1) * (-0.25 * dt * dt));
    putF(0, 24, (R1 * Ra1) * (0.25 * dt * dt * dt));
    putF(3, 3, Rwx * (-dt));
    putF(3, 24, I3 * (-1.0 * dt));
    putF(6, 3, kappa_1);
    putF(6, 21, (R0 + R1) * (-0.5 * dt));
    putF(6, 24, (R1 * Ra1) * (0.5 * dt * dt));
    const m3 VpG = (R1 * Ra1) * (-0.25 * dt * dt * 0.5 * dt);
    putV(0, 0, R0 * (0.25 * dt * dt)); putV(0, 3, VpG); putV(0, 6, R1 * (0.25 * dt * dt)); putV(0, 9, VpG);
    putV(3, 3, I3 * (0.5 * dt)); putV(3, 9, I3 * (0.5 * dt));
    const m3 VvG = (R1 * Ra1) * (-0.5 * dt * 0.5 * dt);
    putV(6, 0, R0 * (0.5 * dt)); putV(6, 3, VvG); putV(6, 6, R1 * (0.5 * dt)); putV(6, 9, VvG);
    for (int j = 0; j < 4; ++j) {
      const int e = 9 + 3 * j;
      const double *l0 = legs + 27 * (2 * j), *l1 = legs + 27 * (2 * j + 1);
      const v3 vi = ld3(l0), vi1 = ld3(l1), p0 = ld3(l0 + 3), p1 = ld3(l1 + 3);
      const m3 hi = R0 * ld_m3_rowmajor(l0 + 6), hi1 = R1 * ld_m3_rowmajor(l1 + 6), Ji = ld_m3_rowmajor(l0 + 15), Ji1 = ld_m3_rowmajor(l1 + 15);
      const v3 gi = -(R0 * ld3(l0 + 24)), gi1 = -(R1 * ld3(l1 + 24));
      putF(e, 3, (R0 * skew(vi)) * (-0.5 * dt) - (R1 * skew(vi1) * kappa_7) * (0.5 * dt));
      putF(e, 24, (R1 * skew(vi1)) * (0.5 * dt * dt) - (R0 * skew(p0) + R1 * skew(p1)) * (0.5 * dt));
      const v3 gsum = (gi + gi1) * (0.5 * dt);
      dF[(e + 0) * 31 + 27 + j] = gsum.x; dF[(e + 1) * 31 + 27 + j] = gsum.y; dF[(e + 2) * 31 + 27 + j] = gsum.z;
      putV(e, 3, (R1 * skew(vi1)) * (-0.25 * dt * dt) + (R0 * skew(p0)) * (0.5 * dt));
      putV(e, 9, (R1 * skew(vi1)) * (-0.25 * dt * dt) + (R1 * skew(p1)) * (0.5 * dt));
      putV(e, 18, hi * (-0.5 * dt));
      putV(e, 21, hi1 * (-0.5 * dt));
      putV(e, 24, (R0 * Rbr * Ji) * (-0.5 * dt));
      putV(e, 27, (R1 * Rbr * Ji1) * (-0.5 * dt));
      putV(e, 30 + 3 * j, I3 * (-dt));
    }
    for (int j = 0; j < 4; ++j) V[(27 + j) * 48 + 42 + j] = -dt;
    putV(21, 12, I3 * (-dt));
    putV(24, 15, I3 * (-dt));
    return;
  }
  static const pb::Tables tab = pb::make_tables();
  double L[pb::PB_TOTAL];
  for (int i = 0; i < pb::PB_TOTAL; ++i) L[i] = 0.0;
  auto putS = [&](int slot, const m3 &A) { for (int q = 0; q < 9; ++q) L[pb::O_POOL + 9 * slot + q] = A.a[q]; };
  putS(pb::S_R0, R0); putS(pb::S_R1, R1); putS(pb::S_K7, kappa_7); putS(pb::S_RA0, Ra0); putS(pb::S_RA1, Ra1); putS(pb::S_RWX, Rwx);
  putS(pb::S_RBR, Rbr); putS(pb::S_I, I3);
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 2; ++e) {
      const double *l = legs + 27 * (2 * j + e);
      double rec[pb::REC_N];
      const m3 skv = skew(ld3(l)), skp = skew(ld3(l + 3));
      for (int q = 0; q < 9; ++q) { rec[q] = skv.a[q]; rec[9 + q] = skp.a[q]; rec[18 + q] = l[6 + q]; rec[27 + q] = l[15 + q]; }
      for (int q = 0; q < 3; ++q) { rec[36 + q] = l[24 + q]; rec[39 + q] = l[q]; }
      for (int r = 0; r < pb::REC_N; ++r) L[pb::record_dest(j, e, r)] = rec[r];
    }
  pb::coefficients(dt, L + pb::O_COEF);
  auto grp_of = [](int lane) { return lane < 63 ? lane / 9 : 7; };
  for (int round = 0; round < 4; ++round)
    for (int lane = 0; lane < 64; ++lane)
      pb::product_entry(tab.prod[8 * round + grp_of(lane)], lane % 9, L);
  for (int lane = 0; lane < 64; ++lane) pb::gvec_entry(lane, L);
  for (int round = 4; round < 6; ++round)
    for (int lane = 0; lane < 64; ++lane)
      pb::product_entry(tab.prod[8 * round + grp_of(lane)], lane % 9, L);
  for (int round = 0; round < pb::N_BLK_ROUNDS; ++round)   // (no round reads what another writes: any order)
    for (int lane = 0; lane < 64; ++lane)
      pb::block_entry(tab.blk[8 * round + grp_of(lane)], lane % 9, L);
  for (int lane = 0; lane < 64; ++lane) pb::tail_entry(lane, dt, L);
  for (int r = 0; r < 32; ++r) {
    for (int k = 0; k < 16; ++k) dF[r * 31 + pb::fk_col(k)] = L[pb::O_FC + r * pb::FCLD + k];
    for (int c = 0; c < 48; ++c) V[r * 48 + c] = L[pb::O_VM + r * pb::VLD + c];
  }
}

// kernels_preint.hip skips the matrix instructions of V N V^T whose operands are structurally zero: the two k-step masks it is compiled with
void hc_v_kstep_masks(unsigned *lo_only, unsigned *hi_only) { *lo_only = pb::V_KSTEPS_ROWS_LO_ONLY; *hi_only = pb::V_KSTEPS_ROWS_HI_ONLY; }
// the same for dF = F - I in its compact column order: the mask and the 16 columns of F in that order
void hc_df_kstep_mask(unsigned *lo_only, int *cols16) { *lo_only = pb::DF_KSTEPS_ROWS_LO_ONLY; for (int k = 0; k < 16; ++k) cols16[k] = pb::fk_col(k); }
}
