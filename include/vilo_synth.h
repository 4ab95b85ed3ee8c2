/* vilo_synth.h — deterministic synthetic sliding windows (workload generator for tests and bench.py).
 * Implements the synthetic configurations of BASELINE.json / SURVEY.md §8(d): an 11-frame (10-KF)
 * window at 15 Hz, trot-gait A1 with diagonal-pair contacts, 500/400 Hz IMU + joint samples, L stereo
 * landmarks with start_frame = i mod 7 tracked to the last frame. Host-only C++ (no GPU, no oracle).
 * Not part of the reference's API: the reference reads these quantities from ROS topics
 * (src/main.cpp:255-393) and its feature manager (src/featureTracker/feature_manager.h:28-78).
 */
#ifndef VILO_SYNTH_H
#define VILO_SYNTH_H
#include <stdint.h>

#include "vilo_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint64_t seed;
  int32_t n_landmarks;     /* 200 (config 2) / 1000 (config 3) */
  int32_t n_start_frames;  /* landmark i starts at frame i mod n_start_frames (7) */
  double imu_rate_hz;      /* 500 (config 2) / 400 (config 3) */
  double frame_rate_hz;    /* 15 */
  double pixel_noise;      /* 0.5 px (divided by focal length) */
  /* initial-state perturbation sigmas */
  double sig_p, sig_theta, sig_v, sig_ba, sig_bg, sig_rho, sig_lambda_rel;
  double lin_offset_ba, lin_offset_bg, lin_offset_rho; /* preintegration linearisation point = initial estimate + offset*N(0,1) */
  int32_t with_prior;      /* 1: synthetic prior over [pose0..9, sb0, lb0, ex0, ex1, td] (n = 86) */
  int32_t pad;
} vilo_synth_params;

void vilo_synth_default_params(vilo_synth_params *p, int config /* 2 or 3 */);

/* Sizes needed for a window generated with p: observations, samples (all intervals, ctor samples included). */
void vilo_synth_sizes(const vilo_synth_params *p, int32_t *n_obs, int32_t *n_samples);

typedef struct {
  /* landmark table / observations (vilo_window_desc layout) */
  int32_t *lm_start_frame; /* [L] */
  int32_t *lm_obs_offset;  /* [L+1] */
  double *obs;             /* [n_obs][11] */
  uint8_t *obs_is_stereo;  /* [n_obs] */
  /* raw sensor samples per interval: offsets [F] into samples (F-1 intervals) */
  vilo_sample *samples;    /* [n_samples] */
  int32_t *sample_offsets; /* [F] */
  double *lin;             /* [F-1][10] ba bg rho linearisation points */
  /* initial states and ground truth (vilo_window_state layouts) */
  double *pose, *speed_bias, *leg_bias, *ex_pose, *td, *inv_depth;
  double *truth_pose, *truth_speed_bias, *truth_leg_bias, *truth_inv_depth;
  /* synthetic prior (caller buffers: x0 [7*40], J0 [96*96], r0 [96]) */
  vilo_prior *prior;
} vilo_synth_out;

int vilo_synth_window(const vilo_config *cfg, const vilo_synth_params *p, vilo_synth_out *out);

/* ---- a continuous sensor stream of one robot (input side of Estimator::processMeasurements, estimator.cpp:400-521) ----
 * Same trajectory, gait, sensor noise and extrinsics as vilo_synth_window; images at frame_rate_hz looking at a static
 * landmark cloud laid out along the path, with a feature-tracker stand-in (tracks persist while the point stays in both
 * fields of view, never re-use an id, at most max_features per image, optical-flow velocity by finite differences). */
typedef struct {
  uint64_t seed;
  double imu_rate_hz, frame_rate_hz, pixel_noise;
  double t0;               /* trajectory time of the first image: different robots of a fleet use different phases */
  int32_t cloud_per_10m;   /* landmarks per 10 m slab of the corridor */
  int32_t max_features;    /* MAX_CNT of the tracker (yaml max_cnt) */
  double drop_prob;        /* per image probability of losing a track */
  double stereo_prob;      /* probability that a visible right-camera match is reported */
} vilo_synth_stream_params;
typedef struct vilo_synth_stream vilo_synth_stream;
void vilo_synth_stream_default_params(vilo_synth_stream_params *p);
vilo_synth_stream *vilo_synth_stream_create(const vilo_config *cfg, const vilo_synth_stream_params *p);
void vilo_synth_stream_destroy(vilo_synth_stream *s);
/* tic [2][3], ric [2][9] row-major, td */
void vilo_synth_stream_extrinsics(const vilo_synth_stream *s, double *tic, double *ric, double *td);
/* Next image: the samples between the previous image and this one (dt as estimator.cpp:456-462 computes it; the first
 * call returns the single sample at t0), the tracked features (obs rows in vilo_window_desc::obs order) and the true
 * state at the image time: truth[20] = p(3) q(xyzw) v(3) ba(3) bg(3) rho(4). Returns 0, or -1 when a buffer is too small. */
int vilo_synth_stream_next(vilo_synth_stream *s, vilo_sample *samples, int max_samples, int *n_samples, int *ids, double *obs11,
                           uint8_t *stereo, int max_features, int *n_features, double *header, double *truth);

#ifdef __cplusplus
}
#endif
#endif
