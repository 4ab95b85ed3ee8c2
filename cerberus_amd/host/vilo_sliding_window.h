// Host-side sliding-window manager (second half): the state bookkeeping either side of the solve, so that
// consecutive frames of one robot — or of a fleet of robots in lockstep, one batched solve per image — run through
// libvilo_gpu.so with the marginalisation prior carried from frame to frame.
//
//   reference (src/estimator/estimator.cpp)                         here
//   processIMULeg :590-653 / processIMU :554-588                     SlidingWindow::processIMULeg   (state propagation + sample buffers;
//                                                                    the preintegration itself runs on the device, vilo_preintegrate)
//   initFirstIMUPose :524-544, initFirstPose :546-552                initFirstIMUPose, initFirstPose
//   processImage :655-846                                            beginImage -> optimizeBatch -> endImage   (processImage = all three)
//   vector2double :848-901, double2vector :903-1003                  vector2double, double2vector (gauge fix through vilo_gauge_fix)
//   optimization :1054-1458                                          optimizeBatch: vilo_optimize_windows[_resident] (solve + gauge fix +
//                                                                    marginalisation on one device batch; by default the prior and the
//                                                                    preintegration records stay in HBM between frames), any number of
//                                                                    windows per call (independent robots share one launch sequence)
//   slideWindow :1460-1678                                           slideWindow (MARGIN_OLD / MARGIN_SECOND_NEW incl. sample-buffer merge)
//   outliersRejection :1741-1798, reprojectionError :1729-1739       outliersRejection
//
// Not restated (out of scope): initFramePoseByPnP (cv::solvePnP) and solveGyroscopeBias during the first
// eleven frames — poses of the start-up window come from IMU propagation alone, for which the gyro-bias alignment is the
// identity. failureDetection() returns false on its first line in the reference and has no counterpart here.
//
// Parity status: the reference's Estimator cannot be compiled in this image (ROS / OpenCV / Ceres), so this file is pinned only
// through its parts: FeatureWindow against the reference's FeatureManager, the solve / marginalise calls against the oracle
// (tests/test_sliding_window.py replays every dumped window through the oracle). The frame-to-frame trajectory is "parity
// unpinned" in the sense of DESIGN.md §2.
#pragma once
#include <string>
#include <vector>

#include "../../include/vilo_gpu.h"
#include "vilo_feature_window.h"

namespace vilo {

struct SlidingWindowOptions {
  int use_leg = 1;             // USE_LEG: IMULegFactor (1) or IMUFactor (0)
  int resident = 1;            // 1 (with streaming preintegration, no dump): the prior and the preintegration records stay on the device
                               //    between frames (vilo_prior_pool / vilo_preint_streams handles, vilo_optimize_windows_resident)
  int streaming_preintegration = 1;   // 1: intervals live on the device and are push_back()ed; 0: re-integrate changed intervals
  int optimize_leg_bias = 1;   // OPTIMIZE_LEG_BIAS (estimator.cpp:1074)
  int estimate_extrinsic = 0;  // ESTIMATE_EXTRINSIC (estimator.cpp:1092)
  int estimate_td = 0;         // ESTIMATE_TD (estimator.cpp:1105)
  FeatureWindowConfig features;
  vilo_solve_opts solve;
  std::string dump_dir;        // non-empty: every optimisation is written there as a VILOWIN1 file (include/vilo_window_io.h)
  SlidingWindowOptions() { vilo_default_solve_opts(&solve); }
};

class SlidingWindow {
 public:
  enum { WS = VILO_WINDOW_SIZE, NF = VILO_MAX_FRAMES };
  enum SolverFlag { INITIAL = 0, NON_LINEAR = 1 };                  // estimator.h:58-62
  enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };   // estimator.h:64-68

  SlidingWindow(vilo_ctx *ctx, const vilo_config &cfg, const SlidingWindowOptions &opt);
  ~SlidingWindow();
  SlidingWindow(const SlidingWindow &) = delete;
  SlidingWindow &operator=(const SlidingWindow &) = delete;
  void clearState();                                                                   // estimator.cpp:24-110
  void setExtrinsics(const double *tic2x3, const double *ric2x9, double td);           // setParameter :112-174
  void initFirstPose(const double p[3], const double R[9]);
  void initFirstIMUPose(const vilo_sample *samples, int n);
  // Device-resident preintegration objects (vilo_preint_streams, include/vilo_gpu.h) for the window's intervals: slot j of the
  // window uses one of the NF objects base_id .. base_id + NF - 1 of `pool`, so an interval is integrated once, sample by sample,
  // however often MARGIN_SECOND_NEW merges the newest interval into it (the reference's push_back, estimator.cpp:1581-1599;
  // without a pool a changed interval is re-integrated from its buffer). Robots of a fleet share one pool, one push per image.
  // The pool's kind must match USE_LEG (vilo_preint_streams_create / _create_imu). Without a call the window creates a pool of
  // its own on first use.
  void attachStreams(vilo_preint_streams *pool, int base_id);
  // The same for the marginalisation prior: this robot uses slots base_slot and base_slot + 1 of `pool` alternately.
  void attachPriorPool(vilo_prior_pool *pool, int base_slot);
  void setInitialVelocity(const double v[3]) { for (int i = 0; i < 3; ++i) Vs[0][i] = v[i]; }   // the reference starts at rest
  void processIMULeg(const vilo_sample &s);   // s.dt as computed at estimator.cpp:456-462

  // One image. n features: ids, obs11 rows in vilo_window_desc::obs order, stereo flags. Returns a vilo_status.
  int processImage(double header, int n, const int *ids, const double *obs11, const uint8_t *stereo);
  // The same in three steps, so that a fleet shares the device work: beginImage on every robot, one optimizeBatch over those
  // that returned true, endImage on every robot.
  bool beginImage(double header, int n, const int *ids, const double *obs11, const uint8_t *stereo);
  static int optimizeBatch(vilo_ctx *ctx, SlidingWindow *const *windows, int n);
  // Between images: push_back() the samples buffered so far into the windows' device-resident preintegration objects — what the
  // reference does with every message (processIMULeg, estimator.cpp:612-632). Call it per message, per N messages or never: an interval
  // pushed in pieces is bitwise the interval pushed at once, so only when the device work happens changes (the image step finds the
  // interval already integrated). One reset and one push launch per pool for the whole fleet.
  static int pushSamples(vilo_ctx *ctx, SlidingWindow *const *windows, int n);
  void endImage();

  void vector2double();
  void outliersRejection(std::vector<int> *remove_ids) const;
  void slideWindow();

  // --- state, named as in estimator.h:139-205 ---
  int frame_count = 0, solver_flag = INITIAL, marginalization_flag = MARGIN_OLD;
  double Ps[NF][3], Vs[NF][3], Rs[NF][9], Bas[NF][3], Bgs[NF][3], Rho[NF][4], Headers[NF];
  double tic[2][3], ric[2][9], td = 0.0;
  double g[3];
  FeatureWindow f_manager;
  int sum_of_back = 0, sum_of_front = 0, open_ex_estimation = 0;
  bool first_imu = false, init_first_pose_flag = false;
  vilo_sample last_sample;   // acc_0, gyr_0, phi_0, dphi_0, c_0

  // para_* (estimator.h:189-196) and what the last optimisation saw / produced
  std::vector<double> para_Pose, para_SpeedBias, para_LegBias, para_Ex_Pose, para_Td, para_Feature;
  vilo_solve_summary last_summary;
  int n_optimizations = 0;
  int priorDim() const;   // n of last_marginalization_info, 0: none
  bool hasPrior() const { return priorDim() > 0; }
  int intervalSamples(int j) const { return (int)buf_[j].size(); }

 private:
  static int pushPending(vilo_ctx *ctx, SlidingWindow *const *ws, int n, bool read_back);
  struct PriorStore {
    vilo_prior p;
    std::vector<double> x0, J0, r0;
    PriorStore();
    void bind() { p.x0 = x0.data(); p.J0 = J0.data(); p.r0 = r0.data(); }
  };
  void double2vector();
  void slideWindowOld();
  void slideWindowNew();
  void startInterval(int j);   // "new IMULegIntegrationBase{acc_0, ..., Bas[j], Bgs[j], rho}" for frame j
  void fillDesc();
  int dump(const vilo_window_state &before) const;

  vilo_ctx *ctx_;
  vilo_config cfg_;
  SlidingWindowOptions opt_;
  // il_pre_integrations[j] / pre_integrations[j]: samples (element 0 = constructor arguments), linearisation point, result
  std::vector<vilo_sample> buf_[NF];
  double lin_[NF][10];
  std::vector<vilo_preint> pre_;
  std::vector<vilo_preint_imu> pre_imu_;
  bool dirty_[NF];
  // streaming preintegration: object id of each slot, samples of buf_[j] (after element 0) already pushed, constructor pending
  vilo_preint_streams *pool_ = nullptr;
  bool own_pool_ = false;
  vilo_prior_pool *ppool_ = nullptr;
  bool own_ppool_ = false;
  int pslot_base_ = 0;
  bool resident_ = false;
  double sum_dt_[NF];   // host copy of each interval's sum_dt (the 10 s rule of estimator.cpp:1118 needs it without a device read)
  int sid_[NF];
  int pushed_[NF];
  bool need_reset_[NF];
  PriorStore prior_[2];
  int cur_prior_ = 0;
  int pending_ = 0;   // 0 none, 1 first optimisation (INITIAL), 2 steady state
  double back_R0_[9], back_P0_[3];
  // boundary structs of the pending optimisation
  vilo_window_desc desc_;
  vilo_window_state state_;
  std::vector<int32_t> lm_start_, lm_off_;
  std::vector<double> obs_;
  std::vector<uint8_t> stereo_;
};

}  // namespace vilo

// C entry points for tests and non-C++ hosts.
extern "C" {
typedef struct {
  int32_t use_leg, optimize_leg_bias, estimate_extrinsic, estimate_td;
  int32_t max_num_iterations, fixed_iterations;
  const char *dump_dir;   // may be NULL
  int32_t streaming_preintegration, resident;
} vilo_sw_options;
void *vilo_sw_create(vilo_ctx *ctx, const vilo_config *cfg, const vilo_sw_options *opt);
void vilo_sw_destroy(void *h);
// share one pool of device-resident preintegration objects among the robots of a fleet: robot k uses ids 11*k .. 11*k + 10
void vilo_sw_attach_streams(void *h, vilo_preint_streams *pool, int base_id);
void vilo_sw_attach_prior_pool(void *h, vilo_prior_pool *pool, int base_slot);   // robot k: slots 2*k, 2*k + 1
void vilo_sw_set_extrinsics(void *h, const double *tic2x3, const double *ric2x9, double td);
void vilo_sw_init_first_pose(void *h, const double *p, const double *R, const double *v /* may be NULL */);
void vilo_sw_init_first_imu_pose(void *h, const vilo_sample *samples, int n);
void vilo_sw_process_samples(void *h, const vilo_sample *samples, int n);
int vilo_sw_push_samples(vilo_ctx *ctx, void *const *hs, int n_windows);   // SlidingWindow::pushSamples
int vilo_sw_parallel_selfcheck(int n, int repeats);   // the fleet's worker pool: 0 if every item of every job ran exactly once
int vilo_sw_process_image(void *h, double header, int n, const int *ids, const double *obs11, const uint8_t *stereo);
// fleet: n_windows robots, robot w has n_feat[w] features starting at feat_offset[w] in the concatenated arrays
int vilo_sw_process_images(vilo_ctx *ctx, void *const *hs, int n_windows, const double *headers, const int *feat_offset, const int *ids,
                           const double *obs11, const uint8_t *stereo);
// flags[6] = frame_count, solver_flag, marginalization_flag, n_optimizations, feature_count, prior_n (0: none)
void vilo_sw_get_state(void *h, int *flags, double *Ps, double *Rs, double *Vs, double *Bas, double *Bgs, double *Rho, double *tic,
                       double *ric, double *td);
int vilo_sw_last_summary(void *h, vilo_solve_summary *out);
}
