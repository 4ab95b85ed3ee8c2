// Device-resident layout of a batch of independent sliding windows and of the per-window solver state.
//
// Camera-side ("f-block") local dimensions of one window (Nc = 11*19 + 13 = 222):
//   P part (80, dense, lives in LDS):  pose_k -> 6k..6k+5 (k = 0..10), ex0 -> 66..71, ex1 -> 72..77, td -> 78, pad 79
//   B part (11 x 13, block tridiagonal): frame k -> 80 + 13k + c, c: v 0..2, ba 3..5, bg 6..8, rho 9..12
// Landmarks (inverse depths) are the eliminated e-blocks (Ceres DENSE_SCHUR ordering).
#pragma once
#include "vilo_internal.hpp"

// state vector layout (doubles) per window: vector2double order (estimator.cpp:848-901)
#define XO_POSE 0
#define XO_SB 77
#define XO_LB 176
#define XO_EX 220
#define XO_TD 234
#define XSTRIDE 240

#define CD_EX0 66
#define CD_EX1 72
#define CD_TD 78
#define CD_B0 80
#define CD_N 224  // 80 + 143 = 223, padded

// prior pre-assembled once per batch into the solver's LDS image: pose block (80 x 81 padded rows), speed/leg-bias
// diagonal blocks (11 x 13 x 13) and the 13 x 80 coupling rows of the frame the prior touches
#define PD_CLD 81
#define PD_C 0
#define PD_AD (80 * PD_CLD)
#define PD_BP (PD_AD + 11 * 169)
#define PD_N (PD_BP + 13 * 80)

// assembled camera-side system of a window as the single-wave solver streams it (k_assemble -> k_solve_wave)
#define VILO_FF_N 36      // doubles of an interval's contact-force filter state (preint_blocks.hpp: O_FF layout)
#define VILO_LEG_REC 42   // doubles of the per-(sample, leg) record of the contact preintegration (preint_blocks.hpp: REC_N)
#define CIMG_N 3840     // pose system: 15 lower 16 x 16 tiles x 4 accumulator registers x 64 lanes
#define TK_N 14208      // k_chain's hand-over: T(k) of 11 frames, 5 tiles x 4 registers x 64 lanes each (14080) + the reduced right-hand side so far (80, padded)
#define TK_V 14080
#define BI_AD 0         // [11][13][13]  diagonal blocks A_kk of the speed / leg-bias part
#define BI_AOT 1859     // [10][13][13]  A_{k+1,k} transposed: [k][dimension of frame k][dimension of frame k + 1]
#define BI_BS 3552      // [11][16][18]  IMU coupling of frame k's dimensions with poses k-1, k, k+1 (rows 13..15 zero)
#define BI_BP 6720      // [16][80]      prior coupling rows of the frame whose speed / leg-bias block the prior touches (rows 13..15 zero)
#define BI_DIAG 8000    // [CD_N]        diagonal of the camera-side Hessian
#define BI_DH2 8224     // [CD_N]        dogleg diagonal dhat^2
#define BI_V 8448       // [CD_N]        v = g / dhat^2
#define BI_SCAL 8672    // [8]           camera-side sums: q = v^T H v, |D^-1 g|^2, max |g|
#define BI_N 8704

#define CONST_LB 1
#define CONST_EX 2
#define CONST_TD 4

struct WinMeta {
  int n_frames, L, n_chunks, use_leg;
  int lm_off;      // first landmark (device order)
  int chunk_off;   // first group chunk
  int const_mask;
  int prior_n;     // 0: no prior
  int gram_off;    // first Gram slot
  int n_gram;
  int prior_nb;
  int pad;         // frame whose speed/leg-bias block the prior touches (-1: none)
  int wave_off;    // first packed visual wave
  int n_waves;
};

// One wave-sized chunk of the landmarks of a window that share a start frame.
struct ChunkMeta {
  int win, s, n, kmax;      // n <= 64 landmarks, kmax = max observations among them
  int lm_off;               // global device-order index of lane 0
  int lm_local;             // index inside the window
  int gram_off;             // global Gram slot of t = 0 (kmax slots)
  int pad;
  long long obs_off;        // (unused: observations are stored per packed wave, see WaveMeta)
  long long flag_off;
};

// One wave of the visual kernels: up to 4 chunks (different start frames) packed side by side, each starting at a lane
// that is a multiple of 8 (the MFMA Gram pass walks 8 landmarks per trip); observations are stored per wave.
struct WaveMeta {
  int win, nseg, n_lanes, kmax;   // n_lanes: multiple of 8, <= 64; kmax: max over the segments
  int seg_chunk[4];               // global chunk index
  int seg_lane0[4];
  long long obs_off;              // doubles: layout [t][11][n_lanes]
  long long flag_off;             // bytes:   layout [t][n_lanes]  bit0 valid, bit1 stereo
};

// Trust-region / dogleg state per window (Ceres 1.14 TrustRegionMinimizer + DoglegStrategy members).
struct SolverState {
  double radius, mu;
  double x_cost, cand_cost, model_cost_change;
  double gnorm2;      // |D^-1 g|^2           (gradient_.squaredNorm())
  double gnnorm2;     // |gauss_newton_step_|^2
  double gdotgn;      // gradient_ . gauss_newton_step_
  double q;           // |J D^-2 g|^2
  double alpha;
  double coef_a, coef_b;   // delta = -a * g / dhat^2 - b * y
  double dogleg_step_norm;
  double gmax;        // max |g_i| (unscaled gradient)
  double x_norm;
  double vis_cost, imu_cost, prior_cost;
  int iter, num_successful, termination, done;
  int need_lin;       // 1: (re)linearise at x before the next step (reuse_ == false)
  int step_valid;
  int scale_ready;    // Jacobi scaling computed (iteration 0)
  int num_invalid;
  int lin_fail;       // Cholesky failed for every mu < max_mu
  int cur;            // which landmark-gradient buffer belongs to the current linearisation
  int pad[2];
  double cost_trace[64];
  double radius_trace[64];
  long long t_start;         // constant-rate device clock (100 MHz) when the solve began: max_solver_time budget
  long long phase_clk[63];   // shader-clock stamps (last linearisation), profiling build only: 0..9 and 16..23 k_solve_wave / k_solve_mid / k_solve_mw8, 10..11 k_chain, 12..15 k_assemble (per wave), 24..27 k_backsub (k_solve_mw8: 24..28 a chain frame), 28..32 k_visual_linearize (packed wave 0), 33..35 k_imu_linearize (factor 0), 36..46 k_assemble, 47..57 k_solve_mw8's Cholesky-80
};

struct BatchDev {
  int W, n_chunks, n_lm, n_gram, n_waves;
  WinMeta *win;
  ChunkMeta *chunk;
  WaveMeta *wave;
  int *wave_order;            // launch order of the packed waves of k_visual_linearize: longest (kmax) first, equal lengths adjacent
  double *obs;
  unsigned char *flags;
  // states
  double *x, *xc, *x0;        // [W][XSTRIDE] current / candidate / initial
  double *lam, *lamc, *lam0;  // [n_lm]
  int *lm_perm;               // device order -> original index within its window
  unsigned char *lm_s;        // [n_lm] start frame of a landmark (ascending inside a window): its coupling rows below pose s are structurally zero
  // landmark-side linearisation
  double *lm_E, *lm_dh2, *lm_y, *lm_scale, *lm_einv;  // [n_lm]
  double *lm_gbuf[2];         // [n_lm] x 2: landmark gradients of the current linearisation (SolverState::cur) and of the candidate's
  double *lm_w;               // per window: [80][L] at 80 * lm_off
  double *lm_part;            // small batches only (else null): [11 frames][2 cameras][21 terms][n_lm] landmark-side terms of the
                              // frame-parallel linearisation (k_visual_linearize_tpar / k_visual_reduce)
  double *gram;               // [n_gram][VILO_GRAM]
  double *chunk_cost;         // [n_waves][VILO_MAX_FRAMES] partial visual cost per (packed wave, frame offset)
  // IMU factors
  PreintPrepared *prep;       // [W][10]
  double *imu_raw;            // [31*39][W*10]   raw [J (31x38) | r], entry-major over the factors; zeros written once, structural non-zeros per linearisation
  double *imu_lin;            // [W][10][31*39]  whitened J (31x38) | whitened r (col 38)
  double *imu_gram;           // [W][10][780]    packed upper triangle of [J | r]^T [J | r]
  double *imu_cost;           // [W][10]
  unsigned char *imu_skip;    // [W][10] 1: no factor for this interval (sum_dt > 10 s, estimator.cpp:1118,1164) or interval beyond the window
  // prior
  double *prior_H, *prior_b0, *prior_c0, *prior_x0;   // [W][96*96], [W][96], [W], [W][280]
  double *prior_dense;        // [W][PD_N]
  double *prior_hd;           // [W][96] H dx at the current point (written by k_accept)
  int *prior_map, *prior_bsize, *prior_bidx, *prior_bxoff, *prior_bstate;  // [W][96], [W][40] x4
  // camera-side vectors [W][CD_N]
  double *cam_g, *cam_dh2, *cam_y, *cam_scale;
  // block scratch
  double *Lk;                 // [W][11][169] M_k = L_k^-1 of the bias chain
  double *TAg;                // [W][11][169] T_A(k) = L_k^-1 A_{k,k-1} (both written by the chain and read back by the back-substitution sweeps)
  double *Cimg;               // [W][3840] assembled pose system: 15 lower 16 x 16 tiles in FP64-MFMA accumulator order (k_assemble_pose)
  double *Tk;                 // [W][TK_N] three-stage solver: k_chain -> k_solve_wave hand-over (coupling rows T(k) in MFMA operand order)
  double *cam_gin;            // [W][CD_N] gradient at the linearisation point (k_assemble)
  double *Bimg;               // [W][BI_N] speed / leg-bias part of the assembled system (BI_* layout)
  SolverState *st;
  int *status;
  // preintegration inputs behind the records (optional: vilo_batch_set_samples) for re-propagation inside the iteration, and what the
  // sqrt_info preparation needs
  const vilo_sample *rp_samples;   // all intervals of all windows, concatenated
  double *rp_ff;                   // [W * 10][VILO_FF_N] contact-force filter of every interval's IMULegIntegrationBase object (contact_sensor_type 2): repropagate() does not reset it
  double *rp_terms;                // [samples][4 legs][VILO_LEG_REC] leg terms of every sample at the current linearisation point (k_repropagate's scratch)
  const int *rp_offsets;           // [W * 10 + 1] interval f integrates samples [rp_offsets[f], rp_offsets[f + 1]) (first = constructor sample)
  void *rp_pre;                    // [W * 10] vilo_preint / vilo_preint_imu records the preparation reads
  int *prep_bad;                   // [W * 10] covariance of the record not positive definite
  int rp_on, leg;
  int compact;                // 1: every window keeps td constant: the solve passes use the compact 16-column visual rows / Gram slots (visual_lin.hpp GK_*)
  int full_regime;            // 1: a sub-batch of a larger call (vilo_run_on_lanes): the kernel set of a full batch whatever this batch's own size — no small assembly, no frame-parallel visual form — so that a window's answer is the one it has in the whole call as ONE batch, bit for bit
  int *lin_cur;               // [W] SolverState::cur as the last linearisation pass of a small batch saw it (visual_reduce_body in k_assemble_s's launch)
  int *win_bad;               // [W] 1: a preintegration covariance of the window has no sqrt_info: the window fails alone (termination FAILURE)
};
