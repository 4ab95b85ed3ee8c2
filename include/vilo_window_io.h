/* vilo_window_io.h — on-disk format of ONE sliding-window problem.
 *
 * Purpose: carry exactly what Estimator::optimization() reads (estimator.cpp:1054-1241) — and, optionally, what it
 * produced — between a machine that has the reference's real Ceres/Eigen/ROS stack and this repository, so that the
 * trust-region trajectory of real Ceres can pin the oracle and the HIP path (the part DESIGN.md calls "parity unpinned").
 * The header is plain C99 with no dependency beyond <stdio.h>, so the dump call can be pasted into the reference
 * (INTEGRATION.md §5 shows the patch after estimator.cpp:1057 and :1241).
 *
 * File layout (little endian, all reals IEEE binary64), version 1:
 *   char    magic[8] = "VILOWIN1"
 *   int32   header[16] = { version, n_frames, n_landmarks, n_obs, use_leg, leg_bias_const, ex_const, td_const,
 *                          has_prior, has_after, marginalization_flag, sizeof(vilo_config), 0, 0, 0, 0 }
 *   vilo_config cfg
 *   double  pose[n_frames*7], speed_bias[n_frames*9], leg_bias[n_frames*4], ex_pose[14], td[1], inv_depth[L]      (before)
 *   int32   lm_start_frame[L], lm_obs_offset[L+1]
 *   double  obs[n_obs*11];  uint8 obs_is_stereo[n_obs], zero padded to a multiple of 8 bytes
 *   vilo_preint preint[n_frames-1]                      (use_leg == 1)   |  vilo_preint_imu preint_imu[n_frames-1]  (use_leg == 0)
 *   if has_prior: int32 n, n_blocks, block_id[40], block_size[40], block_idx[40], valid, 0, 0;
 *                 double x0[sum block_size], J0[n*n], r0[n]
 *   if has_after: the six state arrays again (the reference's result after ceres::Solve + double2vector),
 *                 double ref_summary[4] = { iterations, initial_cost, final_cost, termination }
 */
#ifndef VILO_WINDOW_IO_H
#define VILO_WINDOW_IO_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vilo_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A loaded file: owns every buffer its desc / state structs point to. */
typedef struct {
  vilo_config cfg;
  vilo_window_desc desc;
  vilo_window_state before, after; /* after.* == NULL when the file has no result */
  vilo_prior prior;
  int32_t has_after, marginalization_flag;
  double ref_summary[4];
  void *blocks[24];
  int n_blocks_owned;
} vilo_window_file;

static int vilo__w(FILE *f, const void *p, size_t n) { return fwrite(p, 1, n, f) == n ? 0 : -1; }
static int vilo__r(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }
static void *vilo__own(vilo_window_file *wf, size_t n) {
  void *p = calloc(n ? n : 1, 1);
  if (p && wf->n_blocks_owned < 24) wf->blocks[wf->n_blocks_owned++] = p;
  return p;
}
static int vilo__w_state(FILE *f, const vilo_window_state *s, int F, int L) {
  return vilo__w(f, s->pose, sizeof(double) * 7 * F) | vilo__w(f, s->speed_bias, sizeof(double) * 9 * F) |
         vilo__w(f, s->leg_bias, sizeof(double) * 4 * F) | vilo__w(f, s->ex_pose, sizeof(double) * 14) |
         vilo__w(f, s->td, sizeof(double)) | vilo__w(f, s->inv_depth, sizeof(double) * (size_t)L);
}
static int vilo__r_state(FILE *f, vilo_window_file *wf, vilo_window_state *s, int F, int L) {
  s->pose = (double *)vilo__own(wf, sizeof(double) * 7 * F); s->speed_bias = (double *)vilo__own(wf, sizeof(double) * 9 * F);
  s->leg_bias = (double *)vilo__own(wf, sizeof(double) * 4 * F); s->ex_pose = (double *)vilo__own(wf, sizeof(double) * 14);
  s->td = (double *)vilo__own(wf, sizeof(double)); s->inv_depth = (double *)vilo__own(wf, sizeof(double) * (size_t)(L ? L : 1));
  if (!s->pose || !s->speed_bias || !s->leg_bias || !s->ex_pose || !s->td || !s->inv_depth) return -1;
  return vilo__r(f, s->pose, sizeof(double) * 7 * F) | vilo__r(f, s->speed_bias, sizeof(double) * 9 * F) |
         vilo__r(f, s->leg_bias, sizeof(double) * 4 * F) | vilo__r(f, s->ex_pose, sizeof(double) * 14) |
         vilo__r(f, s->td, sizeof(double)) | vilo__r(f, s->inv_depth, sizeof(double) * (size_t)L);
}

/* Write one window. `after` and `ref_summary` may be NULL (inputs only). Returns 0 on success. */
static int vilo_window_write(const char *path, const vilo_config *cfg, const vilo_window_desc *d, const vilo_window_state *before,
                             const vilo_window_state *after, const double ref_summary[4], int marginalization_flag) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  const int F = d->n_frames, L = d->n_landmarks, has_prior = (d->prior && d->prior->valid) ? 1 : 0;
  int32_t hdr[16] = {1, F, L, d->n_obs, d->use_leg, d->leg_bias_const, d->ex_const, d->td_const, has_prior, after ? 1 : 0,
                     marginalization_flag, (int32_t)sizeof(vilo_config), 0, 0, 0, 0};
  int rc = vilo__w(f, "VILOWIN1", 8) | vilo__w(f, hdr, sizeof hdr) | vilo__w(f, cfg, sizeof *cfg) | vilo__w_state(f, before, F, L);
  rc |= vilo__w(f, d->lm_start_frame, sizeof(int32_t) * (size_t)L) | vilo__w(f, d->lm_obs_offset, sizeof(int32_t) * (size_t)(L + 1));
  rc |= vilo__w(f, d->obs, sizeof(double) * 11 * (size_t)d->n_obs) | vilo__w(f, d->obs_is_stereo, (size_t)d->n_obs);
  { const char zero[8] = {0}; const size_t pad = (8 - ((size_t)d->n_obs & 7)) & 7; rc |= vilo__w(f, zero, pad); }
  if (d->use_leg) rc |= vilo__w(f, d->preint, sizeof(vilo_preint) * (size_t)(F - 1));
  else rc |= vilo__w(f, d->preint_imu, sizeof(vilo_preint_imu) * (size_t)(F - 1));
  if (has_prior) {
    const vilo_prior *p = d->prior;
    int32_t ph[2 + 3 * VILO_MAX_PRIOR_BLOCKS + 3];
    int k, xs = 0;
    ph[0] = p->n; ph[1] = p->n_blocks;
    for (k = 0; k < VILO_MAX_PRIOR_BLOCKS; ++k) { ph[2 + k] = p->block_id[k]; ph[2 + VILO_MAX_PRIOR_BLOCKS + k] = p->block_size[k]; ph[2 + 2 * VILO_MAX_PRIOR_BLOCKS + k] = p->block_idx[k]; }
    ph[2 + 3 * VILO_MAX_PRIOR_BLOCKS] = p->valid; ph[3 + 3 * VILO_MAX_PRIOR_BLOCKS] = 0; ph[4 + 3 * VILO_MAX_PRIOR_BLOCKS] = 0;
    for (k = 0; k < p->n_blocks; ++k) xs += p->block_size[k];
    rc |= vilo__w(f, ph, sizeof ph) | vilo__w(f, p->x0, sizeof(double) * (size_t)xs) | vilo__w(f, p->J0, sizeof(double) * (size_t)p->n * p->n) |
          vilo__w(f, p->r0, sizeof(double) * (size_t)p->n);
  }
  if (after) {
    const double zero4[4] = {0, 0, 0, 0};
    rc |= vilo__w_state(f, after, F, L) | vilo__w(f, ref_summary ? ref_summary : zero4, sizeof(double) * 4);
  }
  if (fclose(f) != 0) rc = -1;
  return rc ? -1 : 0;
}

static void vilo_window_free(vilo_window_file *wf) {
  int i;
  for (i = 0; i < wf->n_blocks_owned; ++i) free(wf->blocks[i]);
  memset(wf, 0, sizeof *wf);
}

/* Read one window; on success wf->desc / wf->before (/ wf->after) are ready for vilo_solve_windows. */
static int vilo_window_read(const char *path, vilo_window_file *wf) {
  FILE *f = fopen(path, "rb");
  char magic[8];
  int32_t hdr[16];
  int rc, F, L;
  long fsz;
  memset(wf, 0, sizeof *wf);
  if (!f) return -1;
  if (fseek(f, 0, SEEK_END) != 0 || (fsz = ftell(f)) < 0 || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return -1; }
  rc = vilo__r(f, magic, 8) | vilo__r(f, hdr, sizeof hdr);
  if (rc || memcmp(magic, "VILOWIN1", 8) != 0 || hdr[0] != 1 || hdr[11] != (int32_t)sizeof(vilo_config)) { fclose(f); return -2; }
  F = hdr[1]; L = hdr[2];
  if (F < 2 || F > VILO_MAX_FRAMES || L < 0 || hdr[3] < 0) { fclose(f); return -2; }
  /* a dump is untrusted input: the counts it states are bounded by what the file can hold (16 bytes per landmark, 89 per observation)
     before anything is allocated for them */
  if ((uint64_t)L * 16u + (uint64_t)hdr[3] * 89u > (uint64_t)fsz) { fclose(f); return -2; }
  wf->desc.n_frames = F; wf->desc.n_landmarks = L; wf->desc.n_obs = hdr[3]; wf->desc.use_leg = hdr[4];
  wf->desc.leg_bias_const = hdr[5]; wf->desc.ex_const = hdr[6]; wf->desc.td_const = hdr[7];
  wf->has_after = hdr[9]; wf->marginalization_flag = hdr[10];
  rc = vilo__r(f, &wf->cfg, sizeof wf->cfg) | vilo__r_state(f, wf, &wf->before, F, L);
  {
    int32_t *sf = (int32_t *)vilo__own(wf, sizeof(int32_t) * (size_t)(L ? L : 1)), *oo = (int32_t *)vilo__own(wf, sizeof(int32_t) * (size_t)(L + 1));
    double *obs = (double *)vilo__own(wf, sizeof(double) * 11 * (size_t)(hdr[3] ? hdr[3] : 1));
    uint8_t *st = (uint8_t *)vilo__own(wf, (size_t)hdr[3] + 8);
    char padb[8];
    if (!sf || !oo || !obs || !st) { fclose(f); vilo_window_free(wf); return -3; }
    rc |= vilo__r(f, sf, sizeof(int32_t) * (size_t)L) | vilo__r(f, oo, sizeof(int32_t) * (size_t)(L + 1)) |
          vilo__r(f, obs, sizeof(double) * 11 * (size_t)hdr[3]) | vilo__r(f, st, (size_t)hdr[3]) | vilo__r(f, padb, (8 - ((size_t)hdr[3] & 7)) & 7);
    wf->desc.lm_start_frame = sf; wf->desc.lm_obs_offset = oo; wf->desc.obs = obs; wf->desc.obs_is_stereo = st;
  }
  if (wf->desc.use_leg) {
    vilo_preint *p = (vilo_preint *)vilo__own(wf, sizeof(vilo_preint) * (size_t)(F - 1));
    if (!p) { fclose(f); vilo_window_free(wf); return -3; }
    rc |= vilo__r(f, p, sizeof(vilo_preint) * (size_t)(F - 1));
    wf->desc.preint = p;
  } else {
    vilo_preint_imu *p = (vilo_preint_imu *)vilo__own(wf, sizeof(vilo_preint_imu) * (size_t)(F - 1));
    if (!p) { fclose(f); vilo_window_free(wf); return -3; }
    rc |= vilo__r(f, p, sizeof(vilo_preint_imu) * (size_t)(F - 1));
    wf->desc.preint_imu = p;
  }
  if (hdr[8]) {
    int32_t ph[2 + 3 * VILO_MAX_PRIOR_BLOCKS + 3];
    int k, xs = 0;
    rc |= vilo__r(f, ph, sizeof ph);
    if (rc || ph[0] < 0 || ph[0] > VILO_MAX_PRIOR_DIM || ph[1] < 0 || ph[1] > VILO_MAX_PRIOR_BLOCKS) { fclose(f); vilo_window_free(wf); return -2; }
    wf->prior.n = ph[0]; wf->prior.n_blocks = ph[1];
    for (k = 0; k < VILO_MAX_PRIOR_BLOCKS; ++k) { wf->prior.block_id[k] = ph[2 + k]; wf->prior.block_size[k] = ph[2 + VILO_MAX_PRIOR_BLOCKS + k]; wf->prior.block_idx[k] = ph[2 + 2 * VILO_MAX_PRIOR_BLOCKS + k]; }
    wf->prior.valid = ph[2 + 3 * VILO_MAX_PRIOR_BLOCKS];
    for (k = 0; k < wf->prior.n_blocks; ++k) {
      if (wf->prior.block_size[k] < 0 || wf->prior.block_size[k] > 16) { fclose(f); vilo_window_free(wf); return -2; }
      xs += wf->prior.block_size[k];
    }
    wf->prior.x0 = (double *)vilo__own(wf, sizeof(double) * (size_t)(xs ? xs : 1));
    wf->prior.J0 = (double *)vilo__own(wf, sizeof(double) * (size_t)wf->prior.n * wf->prior.n + 8);
    wf->prior.r0 = (double *)vilo__own(wf, sizeof(double) * (size_t)wf->prior.n + 8);
    if (!wf->prior.x0 || !wf->prior.J0 || !wf->prior.r0) { fclose(f); vilo_window_free(wf); return -3; }
    rc |= vilo__r(f, wf->prior.x0, sizeof(double) * (size_t)xs) | vilo__r(f, wf->prior.J0, sizeof(double) * (size_t)wf->prior.n * wf->prior.n) |
          vilo__r(f, wf->prior.r0, sizeof(double) * (size_t)wf->prior.n);
    wf->desc.prior = &wf->prior;
  }
  if (wf->has_after) rc |= vilo__r_state(f, wf, &wf->after, F, L) | vilo__r(f, wf->ref_summary, sizeof(double) * 4);
  fclose(f);
  if (rc) { vilo_window_free(wf); return -2; }
  return 0;
}

#ifdef __cplusplus
}
#endif
#endif
