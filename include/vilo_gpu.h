/* vilo_gpu.h — C ABI of the MI355X-native sliding-window VILO solver.
 *
 * Drop-in boundary for the hot path of ShuoYangRobotics/Cerberus:
 *     void Estimator::optimization()          src/estimator/estimator.cpp:1054-1458
 * i.e. per-frame nonlinear least squares over the 11-frame window (IMU-leg contact-preintegration
 * factors, stereo reprojection factors, marginalisation prior; Ceres DENSE_SCHUR + DOGLEG) followed by
 * Schur-complement marginalisation. All entry points take plain pointers and sizes; there are no
 * torch / Eigen / Ceres types in any signature. Everything is FP64, like the reference.
 *
 * Each entry point names the reference interface it replaces (paths relative to the Cerberus tree).
 * Return value: 0 on success, negative vilo_status on error (the reference's factors silently
 * `return true`, imu_leg_factor.cpp:385; numerical blow-ups only ROS_WARN, imu_factor.h:88-93).
 */
#ifndef VILO_GPU_H
#define VILO_GPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VILO_WINDOW_SIZE 10           /* parameters.h:23 */
#define VILO_MAX_FRAMES 11            /* WINDOW_SIZE + 1 */
#define VILO_NUM_OF_F 1000            /* parameters.h:24 */
#define VILO_RESIDUAL_STATE_SIZE 31   /* parameters.h:103 */
#define VILO_NOISE_SIZE 46            /* parameters.h:104 */
#define VILO_MAX_PRIOR_BLOCKS 40
#define VILO_MAX_PRIOR_DIM 96         /* n <= 10*6 + 9 + 4 + 12 + 1 = 86 in the reference */

typedef enum {
  VILO_OK = 0,
  VILO_ERR_NO_DEVICE = -1,      /* no HIP device / extension cannot run: never falls back to CPU */
  VILO_ERR_BAD_ARG = -2,
  VILO_ERR_HIP = -3,
  VILO_ERR_NUMERIC = -4,        /* NaN/Inf or non-PD reduced system after mu escalation */
  VILO_ERR_UNSUPPORTED = -5
} vilo_status;

/* The reference's mutable config globals (src/utils/parameters.cpp:13-74) as one POD. */
typedef struct {
  double acc_n, acc_n_z, acc_w, gyr_n, gyr_w;
  double g_norm;
  double phi_n, dphi_n;
  double rho_c_n, rho_nc_n;
  double v_n_min_xy, v_n_min_z, v_n_min, v_n_max;
  double v_n_force_thres_ratio, v_n_term1_steep, v_n_term2_var_rescale, v_n_term3_distance_rescale;
  int32_t contact_sensor_type;
  int32_t pad0;
  double rho_fix[4][4];   /* per leg [ox, oy, d, lt]   (estimator.cpp:142-163) */
  double p_br[3];         /* estimator.cpp:140 */
  double R_br[9];         /* row-major, estimator.cpp:141 */
  double focal_length;    /* FOCAL_LENGTH, parameters.h:22; Projection*Factor::sqrt_info = f/1.5 I (estimator.cpp:124-126) */
  double huber_delta;     /* ceres::HuberLoss(1.0), estimator.cpp:1062 */
} vilo_config;

/* Values of config/a1_config/hardware_a1_vilo_config.yaml. */
void vilo_default_config(vilo_config *cfg);

/* One sample of IMULegIntegrationBase::push_back (imu_leg_integration_base.h:33-34): 35 doubles. */
typedef struct {
  double dt;
  double acc[3], gyr[3];
  double phi[12], dphi[12];
  double c[4];
} vilo_sample;

/* Public state of an IMULegIntegrationBase (imu_leg_integration_base.h:73-84). */
typedef struct {
  double sum_dt;
  double delta_p[3];
  double delta_q[4];      /* x y z w */
  double delta_v[3];
  double delta_eps[12];
  double lin_ba[3], lin_bg[3], lin_rho[4];
  double jacobian[31 * 31];    /* row-major */
  double covariance[31 * 31];  /* row-major */
} vilo_preint;

/* Public state of an IntegrationBase (integration_base.h:201-220). */
typedef struct {
  double sum_dt;
  double delta_p[3];
  double delta_q[4];
  double delta_v[3];
  double lin_ba[3], lin_bg[3];
  double jacobian[15 * 15];
  double covariance[15 * 15];
} vilo_preint_imu;

/* Parameter-block ids replace the raw addresses MarginalizationInfo keys on
 * (marginalization_factor.cpp:98-117, estimator.cpp:1358-1370): id = kind*16 + index. */
#define VILO_BLK_POSE 0
#define VILO_BLK_SB 1
#define VILO_BLK_LB 2
#define VILO_BLK_EX 3
#define VILO_BLK_TD 4
#define VILO_BLK_FEAT 5

/* MarginalizationInfo's product (marginalization_factor.h:57-82): the prior consumed by
 * MarginalizationFactor::Evaluate (marginalization_factor.cpp:347-395). */
typedef struct {
  int32_t n;                                  /* residuals = kept local dimension */
  int32_t n_blocks;
  int32_t block_id[VILO_MAX_PRIOR_BLOCKS];    /* after addr_shift */
  int32_t block_size[VILO_MAX_PRIOR_BLOCKS];  /* keep_block_size (global) */
  int32_t block_idx[VILO_MAX_PRIOR_BLOCKS];   /* keep_block_idx - m */
  double *x0;                                 /* keep_block_data, concatenated */
  double *J0;                                 /* linearized_jacobians, n x n row-major */
  double *r0;                                 /* linearized_residuals */
  int32_t valid;
  int32_t pad;
} vilo_prior;

/* What Estimator::optimization() reads (estimator.h:139-205 + f_manager.feature), flattened. */
typedef struct {
  int32_t n_frames;     /* frame_count + 1 */
  int32_t n_landmarks;  /* features with used_num >= 4 in list order (feature_manager.cpp:179-195) */
  int32_t n_obs;
  int32_t use_leg;      /* 1: IMULegFactor (estimator.cpp:1114-1159), 0: IMUFactor (:1160-1171) */
  const int32_t *lm_start_frame;   /* [L] FeaturePerId::start_frame */
  const int32_t *lm_obs_offset;    /* [L+1] into obs; a landmark's observations are consecutive frames */
  const double *obs;               /* [n_obs][11] FeaturePerFrame: point3 pointRight3 velocity2 velocityRight2 cur_td */
  const uint8_t *obs_is_stereo;    /* [n_obs] */
  const vilo_preint *preint;          /* [n_frames-1] il_pre_integrations[i+1] */
  const vilo_preint_imu *preint_imu;  /* [n_frames-1] pre_integrations[i+1] (use_leg == 0) */
  const vilo_prior *prior;            /* last_marginalization_info; NULL or !valid: none */
  int32_t leg_bias_const, ex_const, td_const;   /* SetParameterBlockConstant, estimator.cpp:1074-1105 */
  int32_t pad;
} vilo_window_desc;

/* para_Pose / para_SpeedBias / para_LegBias / para_Ex_Pose / para_Td / para_Feature
 * (estimator.h:189-196) in the layouts of vector2double (estimator.cpp:848-901). */
typedef struct {
  double *pose;        /* [n_frames][7]  px py pz qx qy qz qw */
  double *speed_bias;  /* [n_frames][9]  v ba bg */
  double *leg_bias;    /* [n_frames][4]  rho FL FR RL RR */
  double *ex_pose;     /* [2][7] */
  double *td;          /* [1] */
  double *inv_depth;   /* [L] */
} vilo_window_state;

/* ceres::Solver::Options as set at estimator.cpp:1221-1233 plus the Ceres 1.14 defaults in play. */
typedef struct {
  int32_t max_num_iterations;      /* NUM_ITERATIONS = 12 */
  int32_t fixed_iterations;        /* 1: no tolerance-based early exit (reproducible work; bench/parity) */
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
  double min_relative_decrease;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double min_lm_diagonal, max_lm_diagonal;
  int32_t jacobi_scaling;
  int32_t max_solver_time_us;  /* Solver::Options::max_solver_time_in_seconds (estimator.cpp:1226-1233: SOLVER_TIME = 0.1 s, x 0.8 on MARGIN_OLD) as a
                                * device-clock budget in microseconds, checked per window where Ceres checks it (before an iteration starts);
                                * when spent the window ends with termination NO_CONVERGENCE. 0 (default): no budget — a solve takes ~3 ms.
                                * This field took the place of a former `reserved` int: fill the struct with vilo_default_solve_opts() first
                                * (a negative value is rejected with VILO_ERR_BAD_ARG) */
} vilo_solve_opts;
void vilo_default_solve_opts(vilo_solve_opts *o);

/* ceres::Solver::Summary subset. */
typedef struct {
  int32_t iterations, num_successful, termination /* 0 no_convergence, 1 convergence, 2 failure */, pad;
  double initial_cost, final_cost;
  double cost_trace[64];
  double radius_trace[64];
} vilo_solve_summary;

typedef struct vilo_ctx vilo_ctx;      /* one per host thread / GPU; owns device buffers and one HIP stream */
typedef struct vilo_batch vilo_batch;  /* device-resident batch of independent windows */

/* Replaces Estimator::setParameter (estimator.cpp:112-174) for the solver side. device = HIP ordinal. */
int vilo_create(vilo_ctx **ctx, const vilo_config *cfg, int device);
void vilo_destroy(vilo_ctx *ctx);
/* HIP devices this process sees (0 without a usable device): the "host thread per GPU" form of SURVEY 8(e) creates one context per ordinal
 * below it, each used from its own thread (tests/host_check/multi_device_check.cpp). */
int vilo_device_count(void);
const char *vilo_last_error(const vilo_ctx *ctx);

/* ---- ceres::CostFunction-shaped batched factor evaluation (host pointers) ------------------------
 * Semantics of CostFunction::Evaluate(double const* const* parameters, double* residuals, double** jacobians):
 * for factor f of n, parameter block k is params_k[f * size_k .. ]; residuals r[f * num_res ..];
 * jac_k, if non-NULL, receives row-major num_res x size_k per factor (global sizes, 7th pose column = 0).
 * obs: [n][12] = pts_i(3) pts_j(3) vel_i(2) vel_j(2) td_i td_j (Projection*Factor constructor arguments). */
/* ProjectionTwoFrameOneCamFactor::Evaluate <2,7,7,7,1,1>  projectionTwoFrameOneCamFactor.cpp:43-150 */
int vilo_eval_proj2f1c(vilo_ctx *ctx, int n, const double *obs, const double *pose_i, const double *pose_j,
                       const double *ex0, const double *inv_dep, const double *td, double *r, double *J_pose_i,
                       double *J_pose_j, double *J_ex0, double *J_feat, double *J_td);
/* ProjectionTwoFrameTwoCamFactor::Evaluate <2,7,7,7,7,1,1>  projectionTwoFrameTwoCamFactor.cpp:43-166 */
int vilo_eval_proj2f2c(vilo_ctx *ctx, int n, const double *obs, const double *pose_i, const double *pose_j,
                       const double *ex0, const double *ex1, const double *inv_dep, const double *td, double *r,
                       double *J_pose_i, double *J_pose_j, double *J_ex0, double *J_ex1, double *J_feat, double *J_td);
/* ProjectionOneFrameTwoCamFactor::Evaluate <2,7,7,1,1>  projectionOneFrameTwoCamFactor.cpp:42-134 */
int vilo_eval_proj1f2c(vilo_ctx *ctx, int n, const double *obs, const double *ex0, const double *ex1,
                       const double *inv_dep, const double *td, double *r, double *J_ex0, double *J_ex1,
                       double *J_feat, double *J_td);
/* IMULegFactor::Evaluate <31,7,9,4,7,9,4>  imu_leg_factor.cpp:173-386 */
int vilo_eval_imu_leg(vilo_ctx *ctx, int n, const vilo_preint *pre, const double *pose_i, const double *sb_i,
                      const double *lb_i, const double *pose_j, const double *sb_j, const double *lb_j, double *r,
                      double *J_pose_i, double *J_sb_i, double *J_lb_i, double *J_pose_j, double *J_sb_j, double *J_lb_j);
/* IMUFactor::Evaluate <15,7,9,7,9>  imu_factor.h:28-188 */
int vilo_eval_imu(vilo_ctx *ctx, int n, const vilo_preint_imu *pre, const double *pose_i, const double *sb_i,
                  const double *pose_j, const double *sb_j, double *r, double *J_pose_i, double *J_sb_i,
                  double *J_pose_j, double *J_sb_j);
/* MarginalizationFactor::Evaluate  marginalization_factor.cpp:347-395. params: concatenated kept blocks in
 * prior order, n_eval evaluations back to back; J (optional): n x sum(block_size) row-major per evaluation. */
int vilo_eval_prior(vilo_ctx *ctx, int n_eval, const vilo_prior *prior, const double *params, double *r, double *J);
/* PoseLocalParameterization::Plus  pose_local_parameterization.cpp:12-27 (batched) */
int vilo_pose_plus(vilo_ctx *ctx, int n, const double *x, const double *delta, double *x_plus_delta);
/* ceres::HuberLoss::Evaluate (used at marginalization_factor.cpp:53) — host-side helper */
void vilo_huber(double delta, double s, double rho[3]);

/* ---- IMULegIntegrationBase(ctor) + push_back/repropagate  imu_leg_integration_base.cpp:7-136 ---------
 * Interval i integrates samples[offsets[i] .. offsets[i+1]): the first sample of the range plays the
 * constructor's (acc_0, gyr_0, phi_0, dphi_0, c_0) (its dt is ignored), the rest are push_back()ed.
 * lin: [n][10] = ba(3) bg(3) rho(4). */
int vilo_preintegrate(vilo_ctx *ctx, int n_intervals, const vilo_sample *samples, const int32_t *offsets,
                      const double *lin, vilo_preint *out);
/* IntegrationBase equivalent (integration_base.h:18-170); lin: [n][6] = ba bg. */
int vilo_preintegrate_imu(vilo_ctx *ctx, int n_intervals, const vilo_sample *samples, const int32_t *offsets,
                          const double *lin, vilo_preint_imu *out);

/* ---- the same objects kept on the device and updated as samples arrive: IMULegIntegrationBase::push_back per IMU/leg
 * message (Estimator::processIMULeg, estimator.cpp:619-626) instead of re-integrating an interval. A pool of n objects;
 * pushing an interval in pieces gives bitwise the result of vilo_preintegrate on the whole interval. ---------------------- */
typedef struct vilo_preint_streams vilo_preint_streams;
int vilo_preint_streams_create(vilo_ctx *ctx, int n, vilo_preint_streams **pool);
/* the same pool of IntegrationBase objects (integration_base.h) for USE_LEG = 0: reset takes lin [n][6] = ba bg, the leg fields
 * of the samples are ignored, vilo_preint_streams_read_imu returns their state */
int vilo_preint_streams_create_imu(vilo_ctx *ctx, int n, vilo_preint_streams **pool);
void vilo_preint_streams_destroy(vilo_ctx *ctx, vilo_preint_streams *pool);
/* new IMULegIntegrationBase{acc_0, gyr_0, phi_0, dphi_0, c_0, ba, bg, rho} (imu_leg_integration_base.cpp:7-42) for the
 * objects ids[0..n): first[k] holds the constructor's measurement, lin [n][10] = ba bg rho. ids must be distinct. */
int vilo_preint_streams_reset(vilo_ctx *ctx, vilo_preint_streams *pool, int n, const int32_t *ids, const vilo_sample *first,
                              const double *lin);
/* push_back (imu_leg_integration_base.cpp:44-60): object ids[k] receives samples[offsets[k] .. offsets[k+1]) in order. */
int vilo_preint_streams_push(vilo_ctx *ctx, vilo_preint_streams *pool, int n, const int32_t *ids, const vilo_sample *samples,
                             const int32_t *offsets);
/* the public state of the objects ids[0..n) (what IMULegFactor reads) */
int vilo_preint_streams_read(vilo_ctx *ctx, vilo_preint_streams *pool, int n, const int32_t *ids, vilo_preint *out);
int vilo_preint_streams_read_imu(vilo_ctx *ctx, vilo_preint_streams *pool, int n, const int32_t *ids, vilo_preint_imu *out);

/* ---- Estimator::optimization(), solve half (estimator.cpp:1054-1245) --------------------------------
 * Synchronous; n_windows = 1 reproduces the reference call. States are updated in place with the
 * solver result (double2vector's gauge fix is vilo_gauge_fix below). */
int vilo_solve_windows(vilo_ctx *ctx, int n_windows, const vilo_window_desc *in, vilo_window_state *inout,
                       const vilo_solve_opts *opts, vilo_solve_summary *out);
/* Many host windows in one vilo_solve_windows / vilo_optimize_windows call (from 2 * sub_windows up, and more than 256 of them with
 * landmarks: a call that as one batch takes the full batch's kernels) go through `lanes` internal contexts of the same device in
 * sub-batches, one host thread per lane, so that packing, the PCIe transfer and the solver work at the same time. The sub-batches run the
 * full batch's kernel set and the whole call's solver form whatever their size: the result is the one batch's bit for bit, and `inout` is
 * written only if every sub-batch came through. Default 4 lanes of 1024 windows (VILO_HOST_PIPELINE="lanes,sub_windows" at vilo_create);
 * lanes < 2 or sub_windows == 0: always one batch. */
int vilo_set_host_pipeline(vilo_ctx *ctx, int lanes, int sub_windows);

/* Device-resident form of the same call, for batches (independent windows: robots / replays / seeds). */
int vilo_batch_create(vilo_ctx *ctx, int n_windows, const vilo_window_desc *in, const vilo_window_state *init,
                      vilo_batch **batch);
int vilo_batch_reset(vilo_ctx *ctx, vilo_batch *batch);  /* restore the uploaded initial states (device-side copy) */
/* sqrt_info = LLT(cov^-1)^T of the batch's preintegration records again (asynchronous): vilo_batch_create runs it once; the reference
 * recomputes it in every IMULegFactor::Evaluate (imu_leg_factor.cpp:197-198), a caller that replays a resident batch can charge it per solve */
int vilo_batch_prepare(vilo_ctx *ctx, vilo_batch *batch);
/* BASELINE configs[2] ("K1 re-propagation of all 10 intervals inside the iteration"): hand the batch the samples behind its IMU-leg
 * records — samples [offsets[10 w + k], offsets[10 w + k + 1]) are interval k of window w, the first one the constructor sample
 * (IMULegIntegrationBase(acc_0, gyr_0, ...), imu_leg_integration_base.cpp:7-42), the rest its push_back()s; n_windows * 10 + 1 offsets,
 * empty ranges for intervals the window does not have. From then on vilo_batch_solve integrates every live interval again
 * (IMULegIntegrationBase::repropagate(Bai, Bgi, rhoi), imu_leg_integration_base.cpp:62-86) at the biases of every point it
 * linearises or evaluates, followed by the sqrt_info of the new covariance, and vilo_batch_marginalize does so at the accepted state.
 * samples == NULL switches it off again. The records the batch was built with are overwritten. */
int vilo_batch_set_samples(vilo_ctx *ctx, vilo_batch *batch, const vilo_sample *samples, const int32_t *offsets);
int vilo_batch_solve(vilo_ctx *ctx, vilo_batch *batch, const vilo_solve_opts *opts);
int vilo_batch_download(vilo_ctx *ctx, vilo_batch *batch, vilo_window_state *out, vilo_solve_summary *summaries);
void vilo_batch_destroy(vilo_ctx *ctx, vilo_batch *batch);
/* GPU time of the last vilo_batch_solve on ctx's stream (HIP events), and per-kernel-group breakdown. */
double vilo_last_solve_ms(const vilo_ctx *ctx);
/* Host wall time inside the last vilo_batch_create on ctx (what handing over HOST windows costs before the first kernel): out_ms[0] total,
 * [1] packing into the device layouts, [2] allocation + upload of observations / states / priors, [3] preintegration records up + their
 * sqrt_info; *bytes_up (optional): bytes moved to the device. vilo_last_download_ms: the same for the last vilo_batch_download. */
int vilo_last_create_ms(const vilo_ctx *ctx, double out_ms[4], double *bytes_up);
double vilo_last_download_ms(const vilo_ctx *ctx);

/* double2vector gauge fix (estimator.cpp:903-957): yaw/position re-anchoring of the solver output. */
int vilo_gauge_fix(vilo_ctx *ctx, int n_windows, const vilo_window_state *before, vilo_window_state *after, int n_frames);

/* ---- marginalisation half (estimator.cpp:1247-1455; MarginalizationInfo::{preMarginalize,marginalize,
 * getParameterBlocks} marginalization_factor.cpp:119-333). mode 0 = MARGIN_OLD, 1 = MARGIN_SECOND_NEW.
 * out->x0/J0/r0 must point at caller buffers of >= 7*VILO_MAX_PRIOR_BLOCKS, VILO_MAX_PRIOR_DIM^2, VILO_MAX_PRIOR_DIM doubles.
 * MARGIN_SECOND_NEW with a prior that does not hold para_Pose[WINDOW_SIZE-1] marginalises nothing and hands the
 * incoming prior back unchanged (estimator.cpp:1379-1380); without any prior out->valid = 0. */
int vilo_marginalize(vilo_ctx *ctx, int n_windows, const vilo_window_desc *in, const vilo_window_state *state,
                     int mode, vilo_prior *out);
/* The same on a batch that is already resident, linearised at its device state (after vilo_batch_solve + vilo_batch_download; `state` is
 * the host copy of that state, `in` the descriptors the batch was created from). modes[w]: 0, 1 as above, < 0: leave window w alone.
 * With vilo_batch_set_samples in force the intervals are integrated again at the accepted state first. */
int vilo_batch_marginalize(vilo_ctx *ctx, vilo_batch *batch, int n_windows, const vilo_window_desc *in, const vilo_window_state *state,
                           const int *modes, vilo_prior *out);

/* ---- Estimator::optimization() as ONE call (estimator.cpp:1054-1458): solve, double2vector gauge fix, marginalisation
 * linearised at that result, all on one device-resident batch (one packing, no host round trip between the halves).
 * inout: states, replaced by the gauge-fixed result. marginalization_flag: [n_windows] 0 MARGIN_OLD / 1 MARGIN_SECOND_NEW
 * (estimator.h:64-68), or NULL to skip the marginalisation; windows with n_frames < WINDOW_SIZE + 1 are not marginalised
 * (estimator.cpp:1243) and their next_prior entry is left untouched. next_prior: [n_windows], buffers as for vilo_marginalize. */
int vilo_optimize_windows(vilo_ctx *ctx, int n_windows, const vilo_window_desc *in, vilo_window_state *inout,
                          const vilo_solve_opts *opts, const int *marginalization_flag, vilo_prior *next_prior,
                          vilo_solve_summary *summaries);

/* ---- device-resident hand-over between frames ("a device-resident prior") ----------------------------
 * last_marginalization_info objects kept in HBM: slot = {n, kept blocks, keep_block_data on the host; linearized_jacobians and
 * linearized_residuals on the device}. A window then names the slot its prior comes from and the slot the next prior goes to,
 * and the preintegration objects (vilo_preint_streams) its IMU factors read, instead of carrying 74 KB of J0 and 156 KB of
 * records through host memory in both directions every frame. */
typedef struct vilo_prior_pool vilo_prior_pool;
int vilo_prior_pool_create(vilo_ctx *ctx, int n_slots, vilo_prior_pool **pool);
void vilo_prior_pool_destroy(vilo_ctx *ctx, vilo_prior_pool *pool);
int vilo_prior_pool_upload(vilo_ctx *ctx, vilo_prior_pool *pool, int slot, const vilo_prior *prior);   /* NULL / !valid: empties the slot */
int vilo_prior_pool_download(vilo_ctx *ctx, vilo_prior_pool *pool, int slot, vilo_prior *out);        /* buffers as for vilo_marginalize */
int vilo_prior_pool_dim(const vilo_prior_pool *pool, int slot);   /* n of the slot, 0: no prior */

typedef struct {
  vilo_preint_streams *preint_pool;   /* NULL: the window's vilo_window_desc::preint records are used */
  const int32_t *preint_ids;          /* [n_frames - 1] object of interval k (frames k -> k+1) */
  const double *preint_sum_dt;        /* [n_frames - 1] their sum_dt, host side (intervals above 10 s carry no factor, estimator.cpp:1118) */
  vilo_prior_pool *prior_pool;        /* NULL: vilo_window_desc::prior / the next_prior argument are used */
  int32_t prior_slot;                 /* < 0: no prior */
  int32_t next_prior_slot;            /* where vilo_optimize_windows_resident leaves the new last_marginalization_info */
} vilo_resident_refs;

/* vilo_optimize_windows with per-window device handles (refs[w]); next_prior may be NULL when every window names a prior pool. */
int vilo_optimize_windows_resident(vilo_ctx *ctx, int n_windows, const vilo_window_desc *in, const vilo_resident_refs *refs,
                                   vilo_window_state *inout, const vilo_solve_opts *opts, const int *marginalization_flag,
                                   vilo_prior *next_prior, vilo_solve_summary *summaries);

/* GPU time (HIP events on ctx's stream) of the kernels of the last vilo_marginalize: linearisation + marginalisation. */
double vilo_last_marginalize_ms(const vilo_ctx *ctx);

/* ---- measurement / test hooks (no counterpart in the reference) -------------------------------------- */
/* Windows of the last vilo_marginalize whose Amm was not certified positive definite beyond eps = 1e-8 and therefore went
 * through the eigen-thresholded pseudo-inverse of the full Amm (marginalization_factor.cpp:281-286) instead of block
 * elimination. The environment variable VILO_MARG_GENERAL=1 forces every window down that path. */
int vilo_debug_marg_general_count(const vilo_ctx *ctx);
/* Per-kernel GPU time of the solve pipeline, HIP events on ctx's stream. kinds: see vilo_kernel_name(). */
/* How sqrt_info = LLT(covariance.inverse()).matrixL().transpose() (imu_leg_factor.cpp:197-198, imu_factor.h) is computed for the batches and
 * factor evaluations that follow. 0 (default): Cholesky of the index-reversed covariance and a triangular inverse — the same matrix without
 * forming the inverse. 1: the reference's route literally (inverse by pivoted Gauss-Jordan elimination, then LLT). Both give the exact
 * matrix to a few 1e-15 row by row (the covariance's condition number of 1e13..1e14 is units: ~15 after diagonal equilibration), and the
 * whitened residuals / Jacobians of either agree with the compiled reference's to 1e-13 (tests/test_golden.py); mode 1 exists so that the
 * reference's formula is also there as written. */
int vilo_set_sqrt_info_mode(vilo_ctx *ctx, int mode);
/* Form of the prior's square root that vilo_marginalize / vilo_batch_marginalize / vilo_optimize_windows* leave (per context).
 * VILO_PRIOR_EIGEN (default): J0 = sqrt(S) V^T, r0 = S^-1/2 V^T b over the eigenpairs with S > 1e-8, what MarginalizationInfo::marginalize
 * writes (marginalization_factor.cpp:297-305; rows of J0 mutually orthogonal, defined up to order and sign).
 * VILO_PRIOR_FACTOR: J0 = X^T, r0 = X_p^-1 b_p for the diagonally pivoted Cholesky factor X X^T = A' (n x r; a semi-definite A' — the
 * gauge directions — gives r < n, the rest of J0 is zero rows like the eigen form's dropped ones; X_p: its pivot rows), wherever the
 * device certifies that X X^T has no eigenvalue in (0, 1e-8]  (1 / |X_p^-1|_F^2 > 1e-8) — an orthogonal transformation of the eigen
 * form: J0^T J0, J0^T r0 and |r0|^2, which is all MarginalizationFactor::Evaluate's contribution to a solve depends on, are the same to
 * rounding, but the ROWS of J0 (and the residual vector of the factor) are in another basis. A window that cannot be certified gets the
 * eigen form. For callers that never look at J0 itself (a replay, a resident prior pool): a single window's marginalisation takes a
 * quarter of the time. */
#define VILO_PRIOR_EIGEN 0
#define VILO_PRIOR_FACTOR 1
int vilo_set_prior_form(vilo_ctx *ctx, int form);
/* Solver form of the batches this context solves from here on. VILO_SOLVER_AUTO (default) picks by batch size: eight waves per window
 * (one workgroup per window, its waves in fixed roles) up to 512 windows (VILO_MW8_MAX_WINDOWS; one per CU, a second round from 257 on), one wave up to 1024, the single wave in three kernels beyond. All
 * forms restate the same algorithm; the eight-wave form eliminates in another order and agrees with the single wave to rounding (1e-9 on
 * well-conditioned windows), SPLIT and WAVE agree bitwise. Pin a form when a window must get the same answer whatever the size of the
 * batch it shares. The environment variable VILO_SOLVER (wave | mw8 | split) only sets the default a context is created with. */
#define VILO_SOLVER_AUTO (-1)
#define VILO_SOLVER_WAVE 0
#define VILO_SOLVER_SPLIT 3
#define VILO_SOLVER_MW8 4
int vilo_set_solver_form(vilo_ctx *ctx, int form);
int vilo_get_solver_form(const vilo_ctx *ctx);
/* Batches created from here on may (1, default) or may not (0) use the compact 16-column visual rows / Gram slots (they apply while td
 * is a constant block in every window of the batch); the two row forms agree to rounding. VILO_NO_COMPACT=1 sets the default to 0. */
int vilo_set_compact_rows(vilo_ctx *ctx, int on);
/* Per-kernel HIP-event timing of the solve loop on the context's stream: 0 off (resident batches replay a hipGraph), 1 every kernel,
 * 2 + k only kernel kind k (vilo_kernel_name(k)). Resets the accumulated times. */
void vilo_set_profiling(vilo_ctx *ctx, int on);
int vilo_get_kernel_times(const vilo_ctx *ctx, double *ms, long long *launches, int n);
/* Test hook: DoglegStrategy's mu at the start of the following solves (default 1e-8 = Ceres' min_mu), to resume a solve from a state
 * (x, radius, mu) reached elsewhere: single steps are compared with the oracle this way (tests/test_branches.py). */
int vilo_debug_set_initial_mu(vilo_ctx *ctx, double mu);
const char *vilo_kernel_name(int kind);
/* Copy an internal device array of one window to the host (tests localise parity failures with it). */
int vilo_debug_fetch(vilo_ctx *ctx, vilo_batch *batch, int what, int win, double *out, int max_n);
/* Streams n doubles (8 B per lane) `reps` times: known byte count to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE. */
int vilo_debug_calib_copy(vilo_ctx *ctx, size_t n_doubles, int reps);

#ifdef __cplusplus
}
#endif
#endif /* VILO_GPU_H */
