/* ORACLE — TEST INFRASTRUCTURE ONLY. Nothing under cerberus_amd/ (the product) may include,
 * link, import or execute this code; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker / reported CPU baseline.
 *
 * CPU (FP64, scalar) restatement of the optimisation hot path of ShuoYangRobotics/Cerberus:
 *   Estimator::optimization()   /root/reference/src/estimator/estimator.cpp:1054-1458
 * and the factor library it drives (src/factor/, src/legKinematics/). Each function cites
 * the reference lines it follows. The trust-region solver restates Ceres Solver 1.14.0
 * (third-party, not vendored in the reference; pinned at .devcontainer/Dockerfile:69-83).
 *
 * PARITY STATUS: the reference ships no golden vectors / known-answer tests for this path
 * (SURVEY.md §8c). Factor-level math (kinematics, both preintegrations, five factors, pose plus,
 * Huber corrector, marginalisation Schur complement + prior factor) is PINNED against the
 * reference's own sources compiled from /root/reference (oracle/_ref/libref.so, built by
 * oracle/ref_build/; tests/test_oracle_vs_reference.py) and against the outputs frozen from that
 * build (tests/golden/reference_vectors.npz; tests/test_golden.py). The Ceres trust-region
 * trajectory (orc_solve_window) is "PARITY UNPINNED": restated from the published Ceres 1.14
 * algorithm only, no Ceres build is available here. (It has one independent check: the HIP path implements the same
 * algorithm separately, and replaying a sensor stream through both found — and fixed — a deviation of THIS file from
 * Ceres after rejected steps; tests/golden/seq_window_rejected_steps.vwin keeps that case.) The force-based contact model
 * (contact_sensor_type 2) of the preintegration is pinned against the reference too.
 */
#ifndef VILO_ORACLE_H
#define VILO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Global configuration of the reference (src/utils/parameters.cpp:13-74) as a POD. */
typedef struct {
  double acc_n, acc_n_z, acc_w, gyr_n, gyr_w; /* parameters.cpp:124-128 */
  double g_norm;                              /* G = (0,0,g_norm), parameters.cpp:21,130 */
  double phi_n, dphi_n;                       /* joint_angle_n, joint_velocity_n */
  double rho_c_n, rho_nc_n;                   /* leg_bias_c_n, leg_bias_nc_n */
  double v_n_min_xy, v_n_min_z, v_n_min, v_n_max;
  double v_n_force_thres_ratio, v_n_term1_steep, v_n_term2_var_rescale, v_n_term3_distance_rescale;
  int32_t contact_sensor_type;                /* 0/1 flag, 2 force model */
  int32_t pad0;
  double rho_fix[4][4];                       /* per leg [ox, oy, d, lt], estimator.cpp:142-163 */
  double p_br[3];                             /* estimator.cpp:140 */
  double R_br[9];                             /* row-major, estimator.cpp:141 */
  double focal_length;                        /* 460, parameters.h:22; visual sqrt_info = f/1.5 */
  double huber_delta;                         /* 1.0, estimator.cpp:1062 */
} orc_config;

/* One sensor sample pushed into the preintegration (imu_leg_integration_base.h:33-34). */
typedef struct {
  double dt;
  double acc[3], gyr[3];
  double phi[12], dphi[12];
  double c[4];
} orc_sample; /* 35 doubles */

/* State of an IMULegIntegrationBase after propagation (imu_leg_integration_base.h:73-84). */
typedef struct {
  double sum_dt;
  double delta_p[3];
  double delta_q[4]; /* x y z w */
  double delta_v[3];
  double delta_eps[12];
  double lin_ba[3], lin_bg[3], lin_rho[4];
  double jacobian[31 * 31];   /* row-major */
  double covariance[31 * 31]; /* row-major */
} orc_preint;

/* State of an IntegrationBase (integration_base.h:201-220). */
typedef struct {
  double sum_dt;
  double delta_p[3];
  double delta_q[4]; /* x y z w */
  double delta_v[3];
  double lin_ba[3], lin_bg[3];
  double jacobian[15 * 15];
  double covariance[15 * 15];
} orc_preint_imu;

void orc_default_config(orc_config *cfg); /* config/a1_config/hardware_a1_vilo_config.yaml */

/* A1Kinematics (src/legKinematics/A1Kinematics.cpp:7-221); jac / dJ_* are column-major like Eigen. */
void orc_fk(const double q[3], double lc, const double rho_fix[4], double p[3]);
void orc_jac(const double q[3], double lc, const double rho_fix[4], double J[9]);
void orc_dfk_drho(const double q[3], double lc, const double rho_fix[4], double d[3]);
void orc_dJ_dq(const double q[3], double lc, const double rho_fix[4], double d[27]);
void orc_dJ_drho(const double q[3], double lc, const double rho_fix[4], double d[9]);

/* Preintegration: constructor(first sample's acc/gyr/phi/dphi/c taken from s0) + push_back over
 * samples[0..n) (imu_leg_integration_base.cpp:7-136). s0->dt is ignored. */
void orc_preintegrate_imu_leg(const orc_config *cfg, const orc_sample *s0, const orc_sample *samples, int n,
                              const double ba[3], const double bg[3], const double rho[4], orc_preint *out);
/* repropagate() on an object that integrated its samples before (imu_leg_integration_base.cpp:62-86): as orc_preintegrate_imu_leg, but the
 * contact-force filter of contact_sensor_type 2 (which repropagate does not reset) starts from ff and its final state comes back in ff:
 * min[4], max[4], var[4], window[4][5], window_idx[4] as 36 doubles; all zero = the constructor's state. */
void orc_preintegrate_imu_leg_ff(const orc_config *cfg, const orc_sample *s0, const orc_sample *samples, int n,
                                 const double ba[3], const double bg[3], const double rho[4], double ff[36], orc_preint *out);
void orc_preintegrate_imu(const orc_config *cfg, const orc_sample *s0, const orc_sample *samples, int n,
                          const double ba[3], const double bg[3], orc_preint_imu *out);
/* One midPointIntegration step's F (31x31) and V (31x46), row-major, for FD checks
 * (imu_leg_integration_base.cpp:376-468). */
void orc_imu_leg_step_FV(const orc_config *cfg, const orc_sample *s0, const orc_sample *s1, const double delta_q[4],
                         const double ba[3], const double bg[3], const double rho[4], double *F, double *V);

/* sqrt_info = LLT(cov^-1).matrixL()^T (imu_leg_factor.cpp:197-198). mode 0: UL route (no explicit
 * inverse); mode 1: reference-faithful (LU inverse, then LLT). Returns 0 on success. */
int orc_sqrt_info(const double *cov, int n, int mode, double *sqrt_info);

/* ceres::CostFunction-shaped evaluations. parameters[k] -> k-th block (global size); jacobians may be
 * NULL, jacobians[k] may be NULL; each is row-major num_residuals x global_size_k. */
void orc_eval_imu_leg(const orc_config *cfg, const orc_preint *pre, const double *const *parameters,
                      double *residuals, double **jacobians); /* imu_leg_factor.cpp:173-386 */
void orc_eval_imu(const orc_config *cfg, const orc_preint_imu *pre, const double *const *parameters,
                  double *residuals, double **jacobians); /* imu_factor.h:28-188 */
/* obs12 = pts_i(3), pts_j(3), vel_i(2), vel_j(2), td_i, td_j */
void orc_eval_proj2f1c(const orc_config *cfg, const double obs12[12], const double *const *parameters,
                       double *residuals, double **jacobians); /* projectionTwoFrameOneCamFactor.cpp:43-150 */
void orc_eval_proj2f2c(const orc_config *cfg, const double obs12[12], const double *const *parameters,
                       double *residuals, double **jacobians); /* projectionTwoFrameTwoCamFactor.cpp:43-166 */
void orc_eval_proj1f2c(const orc_config *cfg, const double obs12[12], const double *const *parameters,
                       double *residuals, double **jacobians); /* projectionOneFrameTwoCamFactor.cpp:42-134 */
void orc_pose_plus(const double x[7], const double delta[6], double out[7]); /* pose_local_parameterization.cpp:12-27 */
void orc_huber(double delta, double s, double rho[3]);                       /* ceres::HuberLoss::Evaluate */

/* ---- Marginalisation prior (marginalization_factor.h:57-82) with integer block ids ---- */
/* block id encoding: kind*16 + index; kinds: 0 pose[i], 1 speedbias[i], 2 legbias[i], 3 ex_pose[c], 4 td, 5 feature[k] (k may exceed 15: id = 5*16+k) */
#define ORC_BLK_POSE 0
#define ORC_BLK_SB 1
#define ORC_BLK_LB 2
#define ORC_BLK_EX 3
#define ORC_BLK_TD 4
#define ORC_BLK_FEAT 5
#define ORC_MAX_PRIOR_BLOCKS 40
typedef struct {
  int32_t n;        /* residual count = kept local dimension */
  int32_t n_blocks; /* kept blocks */
  int32_t block_id[ORC_MAX_PRIOR_BLOCKS];   /* after addr_shift */
  int32_t block_size[ORC_MAX_PRIOR_BLOCKS]; /* global size */
  int32_t block_idx[ORC_MAX_PRIOR_BLOCKS];  /* local offset (keep_block_idx - m) */
  double *x0;       /* concatenated global-size snapshots (keep_block_data) */
  double *J0;       /* n x n row-major linearized_jacobians */
  double *r0;       /* n linearized_residuals */
  int32_t valid;
  int32_t pad;
} orc_prior;

/* MarginalizationFactor::Evaluate (marginalization_factor.cpp:347-395). */
void orc_eval_prior(const orc_prior *prior, const double *const *parameters, double *residuals, double **jacobians);

/* ---- Window description (what Estimator::optimization reads) ---- */
typedef struct {
  int32_t n_frames;    /* frame_count + 1 (11 when the window is full) */
  int32_t n_landmarks; /* features with used_num >= 4, list order (feature_manager.cpp:179-195) */
  int32_t n_obs;
  int32_t use_leg;     /* 1: IMULegFactor, 0: IMUFactor (estimator.cpp:1114-1171) */
  const int32_t *lm_start_frame; /* [L] */
  const int32_t *lm_obs_offset;  /* [L+1] */
  const double *obs;             /* [n_obs][11]: point3, pointRight3, velocity2, velocityRight2, cur_td */
  const uint8_t *obs_is_stereo;  /* [n_obs] */
  const orc_preint *preint;          /* [n_frames-1], interval (i,i+1) at index i; use_leg==1 */
  const orc_preint_imu *preint_imu;  /* [n_frames-1]; use_leg==0 */
  const orc_prior *prior;            /* may be NULL / !valid */
  int32_t leg_bias_const, ex_const, td_const; /* estimator.cpp:1074-1105 */
  int32_t pad;
} orc_window;

typedef struct {
  double *pose;       /* [n_frames][7] */
  double *speed_bias; /* [n_frames][9] */
  double *leg_bias;   /* [n_frames][4] */
  double *ex_pose;    /* [2][7] */
  double *td;         /* [1] */
  double *inv_depth;  /* [L] */
} orc_state;

typedef struct {
  int32_t max_num_iterations;    /* 12 */
  int32_t fixed_iterations;      /* 1: disable tolerance-based termination (bench / parity mode) */
  double initial_trust_region_radius; /* 1e4 */
  double max_trust_region_radius;     /* 1e16 */
  double min_trust_region_radius;     /* 1e-32 */
  double min_relative_decrease;       /* 1e-3 */
  double function_tolerance;          /* 1e-6 */
  double gradient_tolerance;          /* 1e-10 */
  double parameter_tolerance;         /* 1e-8 */
  double min_lm_diagonal, max_lm_diagonal; /* 1e-6, 1e32 */
  int32_t jacobi_scaling;             /* 1 */
  int32_t recompute_sqrt_info;        /* 1: reference-faithful (per Evaluate, LU inverse + LLT) */
} orc_solve_opts;

typedef struct {
  int32_t iterations;      /* number of minimizer iterations performed (successful + unsuccessful) */
  int32_t num_successful;
  int32_t termination;     /* 0 no_convergence(max iters), 1 convergence, 2 failure */
  int32_t pad;
  double initial_cost, final_cost;
  double cost_trace[64];   /* cost after each iteration (x_cost_) */
  double radius_trace[64];
} orc_summary;

void orc_default_opts(orc_solve_opts *o);
/* BASELINE configs[2] ("K1 re-propagation inside the iteration"): while samples != NULL every IMULegFactor evaluation of the window
 * solved / marginalised next first integrates its interval again (repropagate, imu_leg_integration_base.cpp:62-86) at the biases
 * of the evaluation point — once per point: an interval evaluated again at the biases it was last integrated at (an accepted candidate's
 * cost, then its Jacobian) keeps its record. Each interval behaves like ONE IMULegIntegrationBase object across those calls: the
 * contact-force filter of contact_sensor_type 2, which repropagate() leaves alone, carries over from pass to pass, the first pass being
 * the object's original integration. The call (also with the same pointers) makes new objects.
 * offsets[i]..offsets[i+1] are interval i's samples, the first being the constructor sample. Not thread safe. */
void orc_set_repropagation(const orc_sample *samples, const int32_t *offsets);
/* Hessian build of orc_marginalize on n threads (the reference: NUM_THREADS = 4 pthreads, marginalization_factor.h:22, .cpp:246-275); default 1. */
void orc_set_marginalize_threads(int n);
/* Test hooks: mu at the start of the following solves (default 1e-8); scalars of the last iteration of the last solve (12 doubles, o_solver.cpp). */
void orc_set_initial_mu(double mu);
void orc_last_step_scalars(double *out12);

/* Cost 1/2 sum rho(|r|^2) at state, and optionally gradient/Hessian pieces in the reduced (camera)
 * ordering used by tests: local layout [frame k: pose6 sb9 lb4]*n_frames, ex0 6, ex1 6, td 1, then landmarks. */
int orc_window_dim(const orc_window *w);
double orc_window_cost(const orc_config *cfg, const orc_window *w, const orc_state *s);
/* Full normal equations at s (no scaling, constant blocks INCLUDED; caller masks): H is dim x dim row-major. */
void orc_window_normal_eq(const orc_config *cfg, const orc_window *w, const orc_state *s, double *H, double *g, double *cost);

/* ceres::Solve restatement: DENSE_SCHUR + traditional DOGLEG (estimator.cpp:1221-1236). In place on s. */
int orc_solve_window(const orc_config *cfg, const orc_window *w, orc_state *s, const orc_solve_opts *o, orc_summary *sum);

/* double2vector gauge fix (estimator.cpp:903-957): s_before = states before the solve (Rs/Ps in window). */
void orc_gauge_fix(const orc_state *before, orc_state *after, int n_frames);
/* Utility::R2ypr / ypr2R (utils/utility.h:83-125), degrees, row-major 3x3 */
void orc_R2ypr(const double R[9], double ypr[3]);
void orc_ypr2R(const double ypr[3], double R[9]);

/* Marginalisation (estimator.cpp:1247-1455 + marginalization_factor.cpp:98-333).
 * mode 0 = MARGIN_OLD, 1 = MARGIN_SECOND_NEW. out arrays must hold >= 128 entries each dimension:
 * out->x0 [>= 7*40], out->J0 [>=128*128], out->r0 [>=128]. Returns 0 ok, 1 when nothing to do / invalid. */
int orc_marginalize(const orc_config *cfg, const orc_window *w, const orc_state *s, int mode, orc_prior *out,
                    double *A_out /* optional (m+n)^2 */, double *b_out, int *m_out);

#ifdef __cplusplus
}
#endif
#endif
