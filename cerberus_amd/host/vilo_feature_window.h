// Host-side feature window (first half): the bookkeeping either side of the solve that turns tracked
// features into the landmark tables of vilo_window_desc and carries depths across the window slide. Same operations and
// results as the reference's FeatureManager (src/featureTracker/feature_manager.cpp), restated on flat storage:
//
//   reference (feature_manager.cpp)                       here
//   addFeatureCheckParallax :52-118                       FeatureWindow::addFrame
//   getFeatureCount :37-49, getDepthVector :179-196        featureCount, depthVector
//   setDepth :142-160, removeFailures :162-171             setDepth, removeFailures
//   clearDepth :173-177, removeOutlier :416-431            clearDepth, removeOutlier
//   triangulatePoint :198-212, triangulate :302-414        triangulate   (stereo / two-view; the multi-view SVD branch of the
//                                                                         reference is unreachable: :347 catches every track
//                                                                         with two observations first)
//   removeBackShiftDepth :433-471, removeBack :473-488     removeBackShiftDepth, removeBack   (MARGIN_OLD slide)
//   removeFront :490-509                                   removeFront                        (MARGIN_SECOND_NEW slide)
//
// Pinned against the reference's own feature_manager.cpp compiled into oracle/_ref/libref.so
// (tests/test_feature_window.py runs both through the same C entry points on random track histories).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/vilo_gpu.h"

namespace vilo {

struct Observation {   // FeaturePerFrame (feature_manager.h:27-59) in the column order of vilo_window_desc::obs
  double point[3], point_right[3], velocity[2], velocity_right[2], cur_td;
  uint8_t is_stereo;
};

struct Track {   // FeaturePerId (feature_manager.h:61-80)
  int feature_id, start_frame;
  double estimated_depth;   // -1: not triangulated yet
  int solve_flag;           // 0 not solved, 1 ok, 2 failed (negative depth)
  std::vector<Observation> obs;
  int endFrame() const { return start_frame + (int)obs.size() - 1; }
};

struct FeatureWindowConfig {
  int window_size = 10;          // WINDOW_SIZE (parameters.h:23)
  double focal_length = 460.0;   // FOCAL_LENGTH
  double min_parallax = 10.0 / 460.0;   // MIN_PARALLAX = keyframe_parallax / FOCAL_LENGTH (parameters.cpp:113-114)
  double init_depth = 5.0;       // INIT_DEPTH (parameters.cpp:107)
  int stereo = 1;                // STEREO
};

class FeatureWindow {
 public:
  explicit FeatureWindow(const FeatureWindowConfig &c = FeatureWindowConfig()) : cfg(c) {}
  void clearState() { tracks.clear(); }
  // One image: n features, ids[i], left observation always, right one when stereo[i] (obs[i] in vilo_window_desc order;
  // cur_td is taken from `td`). Returns the keyframe decision (true: marginalise the oldest frame).
  bool addFrame(int frame_count, int n, const int *ids, const double *obs11, const uint8_t *stereo, double td);
  int featureCount() const;                       // tracks with >= 4 observations
  void depthVector(double *inv_depth) const;      // [featureCount()] inverse depths in list order = para_Feature
  void setDepth(const double *inv_depth);
  void removeFailures();
  void clearDepth();
  void removeOutlier(const int *ids, int n);
  // Ps[k][3], Rs[k][9] row-major body poses of the window, tic[c][3], ric[c][9] extrinsics
  void triangulate(const double *Ps, const double *Rs, const double *tic, const double *ric);
  void removeBackShiftDepth(const double marg_R[9], const double marg_P[3], const double new_R[9], const double new_P[3]);
  void removeBack();
  void removeFront(int frame_count);
  // Landmark tables of the solver boundary for the tracks with >= 4 observations, list order (estimator.cpp:1173-1216).
  // The vectors own the storage the returned desc points to.
  void fill(vilo_window_desc *desc, std::vector<int32_t> *start, std::vector<int32_t> *offset, std::vector<double> *obs,
            std::vector<uint8_t> *stereo) const;

  std::vector<Track> tracks;
  int last_track_num = 0, new_feature_num = 0, long_track_num = 0;
  double last_average_parallax = 0.0;
  FeatureWindowConfig cfg;
};

}  // namespace vilo

// C entry points (tests drive these and the reference's FeatureManager through identical signatures)
extern "C" {
void *vilo_fw_create();
void vilo_fw_destroy(void *h);
int vilo_fw_add_frame(void *h, int frame_count, int n, const int *ids, const double *obs11, const uint8_t *stereo, double td, int *counters3);
int vilo_fw_feature_count(void *h);
void vilo_fw_depth_vector(void *h, double *out);
void vilo_fw_set_depth(void *h, const double *x);
void vilo_fw_remove_failures(void *h);
void vilo_fw_clear_depth(void *h);
void vilo_fw_remove_outlier(void *h, const int *ids, int n);
void vilo_fw_triangulate(void *h, const double *Ps, const double *Rs, const double *tic, const double *ric);
void vilo_fw_remove_back_shift_depth(void *h, const double *marg_R, const double *marg_P, const double *new_R, const double *new_P);
void vilo_fw_remove_back(void *h);
void vilo_fw_remove_front(void *h, int frame_count);
// dump: track_info [n_tracks][4] = id, start_frame, n_obs, solve_flag; depth [n_tracks]; obs [total][11]; stereo [total].
// Returns n_tracks (call with NULLs to size: *total_obs is always set).
int vilo_fw_dump(void *h, int *track_info, double *depth, double *obs11, uint8_t *stereo, int *total_obs);
}
