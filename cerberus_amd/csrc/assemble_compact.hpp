// Visual part of a window's pose system from COMPACT Gram slots (visual_lin.hpp: 16 columns B | theta_i | theta_j | theta_ic |
// theta_ic2 | r per camera): what k_assemble's owner-computes scatter does for the 23-column slots, plus the 3 x 3 transforms that turn
// the B (= d r / d P_i) blocks into the extrinsic-translation blocks once per (chunk, frame) slot.
//
//   slot (start frame s, frame j = s + t):  G = C0 + C1 (upper triangle, tri16), C1B = rows 0..2 of C1 (16 each)
//   Ri' = R_s, Rj' = R_j for t > 0; identity for t = 0 (the one-frame factor's B is reduce ric2^T itself), N0 = Ri' - Rj'
//   pose columns of the slot:   P_s = +B, P_j = -B, theta_s = RI, theta_j = RJ            (none for t = 0: OneFrameTwoCam has no pose blocks)
//   tic  rows:  N0^T C0[B, .] + Ri'^T C1[B, .]  =  N0^T G[B, .] + Rj'^T C1B[.]            (C0[B, .] = G[B, .] - C1B[.])
//   tic2 rows:  -Rj'^T C1B[.]
//   tic x tic:  N0^T C0_BB N0 + Ri'^T C1_BB Ri',   tic x tic2: -Ri'^T C1_BB Rj',   tic2 x tic2: Rj'^T C1_BB Rj'
//
// One owner thread per TARGET entry class (226 of the workgroup's 256 threads), every target of the 80 x 80 image has exactly one
// owner, so the read-modify-writes need no atomics and the sums have a fixed order (batch of N == batch of 1, bitwise):
//   T1  pose_f x pose_f       21   (f = s: sum over t; f = j: per t)         T5  tic  x pose_f   18
//   T2  pose_s x pose_j       36                                             T6  tic2 x pose_f   18
//   T3  pose_f x {theta_ic, theta_ic2, r}  42                                T7  {tic, tic2} x {theta_ic, theta_ic2, r}  42
//   T4  {theta_ic, theta_ic2, r}^2         28                                T8  {tic, tic2}^2   21
// The function is host-compilable (tests/host_check emulates the 256 threads one after the other against a dense accumulation).
#pragma once
#include "visual_lin.hpp"
#define AC_STAGE (11 * VILO_GRAMC)   // doubles: the slots of one chunk

namespace vilo {

#ifndef CD_EX0
#define CD_EX0 66
#define CD_EX1 72
#endif

VD int ac_jidx(int a) { return a < 3 ? a : a + 3; }                     // pose-local dimension a of frame j -> compact column (B with sign -, RJ)
VD double ac_jsign(int a) { return a < 3 ? -1.0 : 1.0; }
VD int ac_kcol(int R) { return R < 3 ? GK_C0 + R : (R < 6 ? GK_C1 + (R - 3) : GK_R); }   // rest index 0..6 -> compact column
VD int ac_restcd(int R) { return R < 3 ? CD_EX0 + 3 + R : CD_EX1 + 3 + (R - 3); }          // rest index 0..5 -> camera dimension
VD int ac_tri(int x, int y) { return x <= y ? tri16(x, y) : tri16(y, x); }

// T8: {tic, tic2} x {tic, tic2}, 21 entries (q: 0..5 tic x tic upper, 6..14 tic x tic2, 15..20 tic2 x tic2 upper):
//   N0^T C0_BB N0 + Ri'^T C1_BB Ri' | -Ri'^T C1_BB Rj' | Rj'^T C1_BB Rj', summed over every frame t of the chunk. The heaviest class (nine
//   3 x 3 terms per slot): three lanes per entry, lane grp of ngrp takes the frames t = grp, grp + ngrp, ...; the caller adds the partial sums
//   in the order grp 0, 1, 2 and applies the total.
VD void ac_t8_decode(int q, int &pa, int &pb, int &ia, int &ib) {
  pa = 0; pb = 0; ia = 0; ib = 0;
  if (q < 6) { int rem = q; while (rem >= 3 - ia) { rem -= 3 - ia; ++ia; } ib = ia + rem; }
  else if (q < 15) { pb = 1; ia = (q - 6) / 3; ib = (q - 6) % 3; }
  else { pa = 1; pb = 1; int rem = q - 15; while (rem >= 3 - ia) { rem -= 3 - ia; ++ia; } ib = ia + rem; }
}
VD double ac_t8_partial(int q, int grp, int ngrp, int s, int km, const double *slots, const double *Rt) {
  int pa, pb, ia, ib;
  ac_t8_decode(q, pa, pb, ia, ib);
  // v = La^T C0 Lb + Ma^T C1 Mb with  (tic, tic): La = N0[:, ia], Lb = N0[:, ib], Ma = Ri'[:, ia], Mb = Ri'[:, ib];
  // (tic, tic2): Ma = Ri'[:, ia], Mb = -Rj'[:, ib];  (tic2, tic2): Ma = Rj'[:, ia], Mb = Rj'[:, ib]  (La = Lb = 0 in the last two)
  const double f0 = (pa == 0 && pb == 0) ? 1.0 : 0.0, sb = (pa == 0 && pb == 1) ? -1.0 : 1.0;
  double sum = 0.0;
  for (int t = grp; t < km; t += ngrp) {
    const double *Gb = slots + t * VILO_GRAMC, *C1B = Gb + VILO_GRAMC_TRI;
    const double *Ri = Rt + 9 * (t ? s : 11), *Rj = Rt + 9 * (t ? s + t : 11);
    const double *RA = pa ? Rj : Ri, *RB = pb ? Rj : Ri;
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // inner sums first: sum_d C[c][d] * right[d]
      double i0 = 0.0, i1 = 0.0;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double c1 = C1B[16 * c + d], c0 = Gb[ac_tri(c, d)] - c1;
        i0 += c0 * (Ri[3 * d + ib] - Rj[3 * d + ib]);
        i1 += c1 * RB[3 * d + ib];
      }
      v += f0 * (Ri[3 * c + ia] - Rj[3 * c + ia]) * i0 + sb * RA[3 * c + ia] * i1;
    }
    sum += v;
  }
  return sum;
}
template <class RMW>
VD void ac_t8_apply(int q, double total, RMW rmw) {
  int pa, pb, ia, ib;
  ac_t8_decode(q, pa, pb, ia, ib);
  rmw((pb ? CD_EX1 : CD_EX0) + ib, (pa ? CD_EX1 : CD_EX0) + ia, total);
}

// T5 + T6: tic (T5) and tic2 (T6) component a x pose_f dimension b of entry q = 6 a + b (18 entries), the second-longest class: two lanes
// per entry, lane par takes the frames t = 1 + par, 3 + par, ... and forms both classes' values from the same reads (T6's are a part of
// T5's). The per-frame targets (pose j = s + t) are applied here; the sums over t (targets in pose s) come back for the caller to add in
// the order par 0, 1 and apply once.
template <class RMW>
VD void ac_t56_partial(int q, int par, int s, int km, const double *slots, const double *Rt, RMW rmw, double &sum5, double &sum6) {
  const int a = q / 6, b = q % 6, xj = ac_jidx(b);
  const double sgj = ac_jsign(b);
  const double *Ri = Rt + 9 * s;
  sum5 = 0.0; sum6 = 0.0;
  for (int t = 1 + par; t < km; t += 2) {
    const double *Gb = slots + t * VILO_GRAMC, *C1B = Gb + VILO_GRAMC_TRI, *Rj = Rt + 9 * (s + t);
    double vs5 = 0.0, vj5 = 0.0, vs6 = 0.0, vj6 = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double rj = Rj[3 * c + a], n0 = Ri[3 * c + a] - rj;
      const double gb = Gb[ac_tri(c, b)], cb = C1B[16 * c + b], gx = Gb[ac_tri(c, xj)], cx = C1B[16 * c + xj];
      vs5 += n0 * gb + rj * cb;
      vj5 += n0 * gx + rj * cx;
      vs6 -= rj * cb;
      vj6 -= rj * cx;
    }
    sum5 += vs5; sum6 += vs6;
    rmw(CD_EX0 + a, 6 * (s + t) + b, sgj * vj5);
    rmw(CD_EX1 + a, 6 * (s + t) + b, sgj * vj6);
  }
}
template <class RMW>
VD void ac_t56_apply(int q, int s, double total5, double total6, RMW rmw) {
  const int a = q / 6, b = q % 6;
  rmw(CD_EX0 + a, 6 * s + b, total5);
  rmw(CD_EX1 + a, 6 * s + b, total6);
}

// One chunk (start frame s, km frames) of one owner thread. slots: the chunk's km slots back to back (k_assemble stages them in LDS with
// coalesced loads: every byte of a slot is fetched from HBM once and the owner threads' scattered reads are LDS reads);  Rt: [12][9]
// rotation matrices of the window's frames (row-major), entry 11 = identity;  rmw(hi, lo, v): image(hi, lo) += v (hi >= lo);
// gadd(cd, v): gradient. Thread ranges are laid out so that a wave of 64 runs at most two of the class bodies:
//   wave 0: T1 [0, 21) T2 [21, 57), then T3 as a second body of lanes [0, 42)   wave 1: T8, three lanes per entry (ac_t8_partial / ac_t8_apply)
//   wave 2: T5 + T6, two lanes per entry [128, 164) (ac_t56_partial / ac_t56_apply), T4 [164, 192)   wave 3: T7 [192, 234)
template <class RMW, class GADD>
VD void assemble_visual_compact_chunk(int tid, int s, int km, const double *slots, const double *Rt, RMW rmw, GADD gadd) {
  if (tid < 21) {
    // T1: pose_f x pose_f, upper (a <= b) of the 6 x 6 block. f = s: [B | RI] x [B | RI] summed over t >= 1; f = j: [-B | RJ] x [-B | RJ] per t
    int a = 0, rem = tid;
    while (rem >= 6 - a) { rem -= 6 - a; ++a; }
    const int b = a + rem;
    const int e1 = ac_tri(a, b), e2 = ac_tri(ac_jidx(a), ac_jidx(b));
    const double sg2 = ac_jsign(a) * ac_jsign(b);
    double sum = 0.0;
#pragma unroll 2
    for (int t = 1; t < km; ++t) {
      sum += slots[t * VILO_GRAMC + e1];
      rmw(6 * (s + t) + b, 6 * (s + t) + a, sg2 * slots[t * VILO_GRAMC + e2]);
    }
    rmw(6 * s + b, 6 * s + a, sum);
  } else if (tid < 57) {
    // T2: pose_s (a) x pose_j (b), the full 6 x 6 block, per t
    const int a = (tid - 21) / 6, b = (tid - 21) % 6;
    const int e = ac_tri(a, ac_jidx(b));
    const double sg = ac_jsign(b);
#pragma unroll 2
    for (int t = 1; t < km; ++t) rmw(6 * (s + t) + b, 6 * s + a, sg * slots[t * VILO_GRAMC + e]);
  }
  if (tid < 42) {
    // T3: pose_f (a) x rest (b: theta_ic 0..2, theta_ic2 3..5, r 6) — a second body of the lanes that own T1 / T2 entries
    const int a = tid / 7, b = tid % 7;
    const int e1 = ac_tri(a, ac_kcol(b)), e2 = ac_tri(ac_jidx(a), ac_kcol(b));
    const double sg2 = ac_jsign(a);
    double sum = 0.0;
#pragma unroll 2
    for (int t = 1; t < km; ++t) {
      sum += slots[t * VILO_GRAMC + e1];
      const double v2 = sg2 * slots[t * VILO_GRAMC + e2];
      if (b == 6) gadd(6 * (s + t) + a, v2);
      else rmw(ac_restcd(b), 6 * (s + t) + a, v2);
    }
    if (b == 6) gadd(6 * s + a, sum);
    else rmw(ac_restcd(b), 6 * s + a, sum);
  }
  if (tid >= 164 && tid < 192) {
    // T4: rest x rest, upper (a <= b) of the 7 x 7 block {theta_ic, theta_ic2, r}, summed over every t
    int a = 0, rem = tid - 164;
    while (rem >= 7 - a) { rem -= 7 - a; ++a; }
    const int b = a + rem;
    const int e = ac_tri(ac_kcol(a), ac_kcol(b));
    double sum = 0.0;
#pragma unroll 2
    for (int t = 0; t < km; ++t) sum += slots[t * VILO_GRAMC + e];
    if (b == 6) { if (a != 6) gadd(ac_restcd(a), sum); }
    else rmw(ac_restcd(b), ac_restcd(a), sum);
  } else if (tid >= 192 && tid < 234) {
    // T7: {tic 0..2, tic2 3..5} (a) x rest (b), summed over every t (identity transforms at t = 0)
    const int a = (tid - 192) / 7, b = (tid - 192) % 7, aa = a % 3, x = ac_kcol(b);
    double sum = 0.0;
#pragma unroll 2
    for (int t = 0; t < km; ++t) {
      const double *Gb = slots + t * VILO_GRAMC, *C1B = Gb + VILO_GRAMC_TRI;
      const double *Ri = Rt + 9 * (t ? s : 11), *Rj = Rt + 9 * (t ? s + t : 11);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double rj = Rj[3 * c + aa];
        const double n0 = (a < 3) ? Ri[3 * c + aa] - rj : 0.0, m1 = (a < 3) ? rj : -rj;
        sum += n0 * Gb[ac_tri(c, x)] + m1 * C1B[16 * c + x];
      }
    }
    const int cd = (a < 3 ? CD_EX0 : CD_EX1) + aa;
    if (b == 6) gadd(cd, sum);
    else { const int rc = ac_restcd(b); rmw(rc > cd ? rc : cd, rc > cd ? cd : rc, sum); }
  }
}


// -------------------------------------------------------------------------------------------------------------------------------------
// The same owner classes for a PASS over at most AC_PASS consecutive frames of a chunk (the full batch's pose assembly,
// kernels_asm_full.hip: its LDS stage holds six slots, so that four workgroups share a CU). slots: the pass's np slots back to back, frames
// t = t0 .. t0 + np - 1 of the chunk, followed at index AC_PASS by a slot of ZEROS. Every body is straight-line: all AC_PASS frames' reads are
// asked for up front, and a frame that is not the body's to take (beyond the pass; frame 0 for the two-frame classes) reads the zero slot —
// its term is an exact 0 added to a valid target, no select and no branch depends on the pass. The targets' positions are spelled out as
// (scalar function of the frame) + (per-lane constant) + at most one 24-bit multiply: cl_pos(R + b, R + a) = [R (R + 1) / 2 + R] + R b +
// [b (b + 1) / 2 + a] with R = 6 (s + t) the same in every lane. Sums over the frames of a chunk reach their target once per pass (pass by
// pass in order; the chunk form adds them once per chunk: rounding, not a different sum).
#define AC_PASS 6
#define AC_ZSLOT AC_PASS   // index of the zero slot behind the pass's slots
// (the compiler's scheduler hoists every LDS read of a straight-line body to its top; where that exceeds the register budget the frames run
//  as a rolled loop, two at a time)
VD int ac_ur(int u, int np) { return u < np ? u : np - 1; }
VD int ac_tri_pos(int r) { return (r * (r + 1)) >> 1; }

// T8 for a pass: lane grp of 3 takes the pass's frames u = grp and grp + 3
VD double ac_t8_pass(int q, int grp, int s, int t0, int np, const double *slots, const double *Rt) {
  int pa, pb, ia, ib;
  ac_t8_decode(q, pa, pb, ia, ib);
  const double f0 = (pa == 0 && pb == 0) ? 1.0 : 0.0, sb = (pa == 0 && pb == 1) ? -1.0 : 1.0;
  double sum = 0.0;
#pragma unroll 1
  for (int j = 0; j < 2; ++j) {
    const int u = grp + 3 * j, t = t0 + ac_ur(u, np);
    const double *Gb = slots + (u < np ? u : AC_ZSLOT) * VILO_GRAMC, *C1B = Gb + VILO_GRAMC_TRI;
    const double *Ri = Rt + 9 * (t ? s : 11), *Rj = Rt + 9 * (t ? s + t : 11);
    const double *RA = pa ? Rj : Ri, *RB = pb ? Rj : Ri;
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double i0 = 0.0, i1 = 0.0;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double c1 = C1B[16 * c + d], c0 = Gb[ac_tri(c, d)] - c1;
        i0 += c0 * (Ri[3 * d + ib] - Rj[3 * d + ib]);
        i1 += c1 * RB[3 * d + ib];
      }
      v += f0 * (Ri[3 * c + ia] - Rj[3 * c + ia]) * i0 + sb * RA[3 * c + ia] * i1;
    }
    sum += v;
  }
  return sum;
}

// T5 + T6 for a pass: lane par of 2 takes the pass's frames u = par, par + 2, par + 4 (those with t >= 1)
template <class RMW>
VD void ac_t56_pass(int q, int par, int s, int t0, int np, const double *slots, const double *Rt, RMW rmw, double &sum5, double &sum6) {
  const int a = q / 6, b = q % 6, xj = ac_jidx(b);
  const double sgj = ac_jsign(b);
  const double *Ri = Rt + 9 * s;
  sum5 = 0.0; sum6 = 0.0;
  double riv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) riv[c] = Ri[3 * c + a];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int u = par + 2 * j, t = t0 + ac_ur(u, np);
    const double *Gb = slots + ((u < np && t0 + u >= 1) ? u : AC_ZSLOT) * VILO_GRAMC, *C1B = Gb + VILO_GRAMC_TRI, *Rj = Rt + 9 * (s + t);
    double vs5 = 0.0, vj5 = 0.0, vs6 = 0.0, vj6 = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double rj = Rj[3 * c + a], n0 = riv[c] - rj;
      const double gb = Gb[ac_tri(c, b)], cb = C1B[16 * c + b], gx = Gb[ac_tri(c, xj)], cx = C1B[16 * c + xj];
      vs5 += n0 * gb + rj * cb;
      vj5 += n0 * gx + rj * cx;
      vs6 -= rj * cb;
      vj6 -= rj * cx;
    }
    sum5 += vs5; sum6 += vs6;
    rmw(CD_EX0 + a, 6 * (s + t) + b, sgj * vj5);
    rmw(CD_EX1 + a, 6 * (s + t) + b, sgj * vj6);
  }
}

// wave 0 of the workgroup: T1 (lanes 0 .. 20), T2 (21 .. 56), then T3 as a second body of lanes 0 .. 41. add_at(pos, v): image[pos] += v
// with pos the position in the packed lower triangle (lin_common.hpp cl_pos).
template <class ADDAT, class GADD>
VD void ac_pass_w0(int tid, int s, int t0, int np, const double *slots, const double *Rt, ADDAT add_at, GADD gadd) {
  (void)Rt;
  // per frame u of the pass (the same in every lane): the slot to read (the zero slot for a frame that is not taken), R = 6 (s + t) of the
  // frame whose targets receive the term (a frame of the chunk in any case), and the scalar parts of the targets' positions
  int slot_u[AC_PASS], R_u[AC_PASS];
#pragma unroll
  for (int u = 0; u < AC_PASS; ++u) {
    const int t = t0 + ac_ur(u, np);
    slot_u[u] = ((u < np && t0 + u >= 1) ? u : AC_ZSLOT) * VILO_GRAMC;
    R_u[u] = 6 * (s + (t > 0 ? t : 1));   // (frame 0 reads zeros: its term lands on frame 1's target, which every lane owns as well)
  }
  const int R0 = 6 * s;
  if (tid < 21) {
    // T1: pose_f x pose_f, upper (a <= b) of the 6 x 6 block. f = s: [B | RI] x [B | RI] summed over t >= 1; f = j: [-B | RJ] x [-B | RJ] per t
    int a = 0, rem = tid;
    while (rem >= 6 - a) { rem -= 6 - a; ++a; }
    const int b = a + rem;
    const int e1 = ac_tri(a, b), e2 = ac_tri(ac_jidx(a), ac_jidx(b)), cb = ac_tri_pos(b) + a;
    const double sg2 = ac_jsign(a) * ac_jsign(b);
    double v1[AC_PASS], v2[AC_PASS];
#pragma unroll
    for (int u = 0; u < AC_PASS; ++u) { v1[u] = slots[slot_u[u] + e1]; v2[u] = slots[slot_u[u] + e2]; }
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < AC_PASS; ++u) {
      sum += v1[u];
      add_at(ac_tri_pos(R_u[u]) + R_u[u] + R_u[u] * b + cb, sg2 * v2[u]);   // (6 (s + t) + b, 6 (s + t) + a)
    }
    add_at(ac_tri_pos(R0) + R0 + R0 * b + cb, sum);
  } else if (tid < 57) {
    // T2: pose_s (a) x pose_j (b), the full 6 x 6 block, per t
    const int a = (tid - 21) / 6, b = (tid - 21) % 6;
    const int e = ac_tri(a, ac_jidx(b)), cb = ac_tri_pos(b) + a;
    const double sg = ac_jsign(b);
    double v[AC_PASS];
#pragma unroll
    for (int u = 0; u < AC_PASS; ++u) v[u] = slots[slot_u[u] + e];
#pragma unroll
    for (int u = 0; u < AC_PASS; ++u) add_at(ac_tri_pos(R_u[u]) + R0 + R_u[u] * b + cb, sg * v[u]);   // (6 (s + t) + b, 6 s + a)
  }
  if (tid < 42) {
    // T3: pose_f (a) x rest (b: theta_ic 0..2, theta_ic2 3..5, r 6) — a second body of the lanes that own T1 / T2 entries
    const int a = tid / 7, b = tid % 7;
    const int e1 = ac_tri(a, ac_kcol(b)), e2 = ac_tri(ac_jidx(a), ac_kcol(b));
    const double sg2 = ac_jsign(a);
    const int rc = (b == 6) ? 0 : ac_restcd(b), cb = ac_tri_pos(rc) + a;
    double v1[AC_PASS], v2[AC_PASS];
#pragma unroll
    for (int u = 0; u < AC_PASS; ++u) { v1[u] = slots[slot_u[u] + e1]; v2[u] = slots[slot_u[u] + e2]; }
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < AC_PASS; ++u) {
      sum += v1[u];
      const double w2 = sg2 * v2[u];
      if (b == 6) gadd(R_u[u] + a, w2);
      else add_at(R_u[u] + cb, w2);   // (restcd(b), 6 (s + t) + a)
    }
    if (b == 6) gadd(R0 + a, sum);
    else add_at(R0 + cb, sum);
  }
}
// T4: {theta_ic, theta_ic2, r}^2, 28 lanes l
template <class RMW, class GADD>
VD void ac_pass_t4(int l, int s, int t0, int np, const double *slots, const double *Rt, RMW rmw, GADD gadd) {
  (void)s; (void)t0; (void)Rt;
  if (l >= 0 && l < 28) {
    int a = 0, rem = l;
    while (rem >= 7 - a) { rem -= 7 - a; ++a; }
    const int b = a + rem;
    const int e = ac_tri(ac_kcol(a), ac_kcol(b));
    double v[AC_PASS];
#pragma unroll
    for (int u = 0; u < AC_PASS; ++u) v[u] = slots[(u < np ? u : AC_ZSLOT) * VILO_GRAMC + e];
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < AC_PASS; ++u) sum += v[u];
    if (b == 6) { if (a != 6) gadd(ac_restcd(a), sum); }
    else rmw(ac_restcd(b), ac_restcd(a), sum);
  }
}
// T7: {tic, tic2} x {theta_ic, theta_ic2, r}, 42 lanes l; two frames' reads in flight at a time (a rolled loop: unrolled, the scheduler asks
// for all six frames' 72 values at once and spills)
template <class RMW, class GADD>
VD void ac_pass_t7(int l, int s, int t0, int np, const double *slots, const double *Rt, RMW rmw, GADD gadd) {
  if (l >= 0 && l < 42) {
    const int a = l / 7, b = l % 7, aa = a % 3, x = ac_kcol(b);
    double sum = 0.0;
#pragma unroll 1
    for (int u0 = 0; u0 < AC_PASS; u0 += 2) {
      double g_[2][3], c_[2][3], ri[2][3], rj[2][3];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int u = u0 + j, t = t0 + ac_ur(u, np);
        const double *Gb = slots + (u < np ? u : AC_ZSLOT) * VILO_GRAMC, *C1B = Gb + VILO_GRAMC_TRI;
        const double *Ri = Rt + 9 * (t ? s : 11), *Rj = Rt + 9 * (t ? s + t : 11);
#pragma unroll
        for (int c = 0; c < 3; ++c) { g_[j][c] = Gb[ac_tri(c, x)]; c_[j][c] = C1B[16 * c + x]; ri[j][c] = Ri[3 * c + aa]; rj[j][c] = Rj[3 * c + aa]; }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double n0 = (a < 3) ? ri[j][c] - rj[j][c] : 0.0, m1 = (a < 3) ? rj[j][c] : -rj[j][c];
          v += n0 * g_[j][c] + m1 * c_[j][c];
        }
        sum += v;
      }
    }
    const int cd = (a < 3 ? CD_EX0 : CD_EX1) + aa;
    if (b == 6) gadd(cd, sum);
    else { const int rc = ac_restcd(b); rmw(rc > cd ? rc : cd, rc > cd ? cd : rc, sum); }
  }
}
// all classes of the generic kind by workgroup thread (the host check's emulation; the kernel calls the per-wave functions)
template <class RMW, class GADD>
VD void assemble_visual_compact_pass(int tid, int s, int t0, int np, const double *slots, const double *Rt, RMW rmw, GADD gadd) {
  // (position in the packed lower triangle back to (row, column): row r starts at r (r + 1) / 2)
  auto add_at = [&](int pos, double v) {
    int r = 0;
    while (ac_tri_pos(r + 1) <= pos) ++r;
    rmw(r, pos - ac_tri_pos(r), v);
  };
  if (tid < 64) ac_pass_w0(tid, s, t0, np, slots, Rt, add_at, gadd);
  ac_pass_t4(tid - 164, s, t0, np, slots, Rt, rmw, gadd);
  ac_pass_t7(tid - 192, s, t0, np, slots, Rt, rmw, gadd);
}

}  // namespace vilo
