// See vilo_feature_window.h. Flat-vector restatement of the reference's feature bookkeeping; no Eigen.
#include "vilo_feature_window.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace vilo {
namespace {

inline void mat3_mul_vec(const double R[9], const double v[3], double out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
inline void mat3T_mul_vec(const double R[9], const double v[3], double out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
}
inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// camera pose as the 3x4 projection [R0^T | -R0^T t0] of body frame `k` and camera `cam` (feature_manager.cpp:312-325)
void camera_projection(const double *Ps, const double *Rs, const double *tic, const double *ric, int k, int cam, double P[12]) {
  double t0[3], R0[9], tmp[3];
  mat3_mul_vec(Rs + 9 * k, tic + 3 * cam, tmp);
  for (int i = 0; i < 3; ++i) t0[i] = Ps[3 * k + i] + tmp[i];
  mat3_mul(Rs + 9 * k, ric + 9 * cam, R0);
  double mt[3];
  mat3T_mul_vec(R0, t0, mt);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) P[4 * i + j] = R0[3 * j + i];
    P[4 * i + 3] = -mt[i];
  }
}

// right singular vector of the smallest singular value of a 4x4 matrix: one-sided Jacobi on its columns
void smallest_right_singular_vector4(const double A_in[16], double v[4]) {
  double U[16], V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::memcpy(U, A_in, sizeof U);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < 4; ++i) { al += U[4 * i + p] * U[4 * i + p]; be += U[4 * i + q] * U[4 * i + q]; ga += U[4 * i + p] * U[4 * i + q]; }
        if (ga == 0.0) continue;
        off = std::max(off, std::fabs(ga) / std::sqrt(al * be + 1e-300));
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 4; ++i) {
          const double up = U[4 * i + p], uq = U[4 * i + q];
          U[4 * i + p] = c * up - s * uq; U[4 * i + q] = s * up + c * uq;
          const double vp = V[4 * i + p], vq = V[4 * i + q];
          V[4 * i + p] = c * vp - s * vq; V[4 * i + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  int best = 0;
  double smin = 1e300;
  for (int j = 0; j < 4; ++j) {
    double s2 = 0;
    for (int i = 0; i < 4; ++i) s2 += U[4 * i + j] * U[4 * i + j];
    if (s2 < smin) { smin = s2; best = j; }
  }
  for (int i = 0; i < 4; ++i) v[i] = V[4 * i + best];
}

// feature_manager.cpp:198-212
void triangulate_point(const double P0[12], const double P1[12], const double p0[2], const double p1[2], double X[3]) {
  double D[16];
  for (int c = 0; c < 4; ++c) {
    D[c] = p0[0] * P0[8 + c] - P0[c];
    D[4 + c] = p0[1] * P0[8 + c] - P0[4 + c];
    D[8 + c] = p1[0] * P1[8 + c] - P1[c];
    D[12 + c] = p1[1] * P1[8 + c] - P1[4 + c];
  }
  double v[4];
  smallest_right_singular_vector4(D, v);
  for (int i = 0; i < 3; ++i) X[i] = v[i] / v[3];
}

}  // namespace

bool FeatureWindow::addFrame(int frame_count, int n, const int *ids, const double *obs11, const uint8_t *stereo, double td) {
  last_track_num = 0; last_average_parallax = 0.0; new_feature_num = 0; long_track_num = 0;
  // the reference walks a std::map keyed by feature id: ascending id order decides the list order of new tracks
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ids[a] < ids[b]; });
  for (int oi : order) {
    Observation o;
    const double *src = obs11 + 11 * oi;
    std::memcpy(o.point, src, sizeof(double) * 3);
    std::memcpy(o.velocity, src + 6, sizeof(double) * 2);
    o.cur_td = td;
    o.is_stereo = stereo[oi] ? 1 : 0;
    for (int k = 0; k < 3; ++k) o.point_right[k] = o.is_stereo ? src[3 + k] : 0.0;
    for (int k = 0; k < 2; ++k) o.velocity_right[k] = o.is_stereo ? src[8 + k] : 0.0;
    auto it = std::find_if(tracks.begin(), tracks.end(), [&](const Track &t) { return t.feature_id == ids[oi]; });
    if (it == tracks.end()) {
      Track t;
      t.feature_id = ids[oi]; t.start_frame = frame_count; t.estimated_depth = -1.0; t.solve_flag = 0;
      t.obs.push_back(o);
      tracks.push_back(t);
      ++new_feature_num;
    } else {
      it->obs.push_back(o);
      ++last_track_num;
      if (it->obs.size() >= 4) ++long_track_num;
    }
  }
  if (frame_count < 2 || last_track_num < 20 || long_track_num < 40 || new_feature_num > 0.5 * last_track_num) return true;
  double parallax_sum = 0.0;
  int parallax_num = 0;
  for (const Track &t : tracks) {
    if (t.start_frame <= frame_count - 2 && t.endFrame() >= frame_count - 1) {
      // compensatedParallax2 (:511-547): displacement on the normalised plane between the two latest keyframe candidates
      const Observation &fi = t.obs[frame_count - 2 - t.start_frame], &fj = t.obs[frame_count - 1 - t.start_frame];
      const double du = fi.point[0] / fi.point[2] - fj.point[0], dv = fi.point[1] / fi.point[2] - fj.point[1];
      parallax_sum += std::max(0.0, std::sqrt(du * du + dv * dv));
      ++parallax_num;
    }
  }
  if (parallax_num == 0) return true;
  last_average_parallax = parallax_sum / parallax_num * cfg.focal_length;
  return parallax_sum / parallax_num >= cfg.min_parallax;
}

int FeatureWindow::featureCount() const {
  int c = 0;
  for (const Track &t : tracks) c += t.obs.size() >= 4 ? 1 : 0;
  return c;
}

void FeatureWindow::depthVector(double *inv_depth) const {
  int k = 0;
  for (const Track &t : tracks)
    if (t.obs.size() >= 4) inv_depth[k++] = 1.0 / t.estimated_depth;
}

void FeatureWindow::setDepth(const double *inv_depth) {
  int k = 0;
  for (Track &t : tracks) {
    if (t.obs.size() < 4) continue;
    t.estimated_depth = 1.0 / inv_depth[k++];
    t.solve_flag = t.estimated_depth < 0 ? 2 : 1;
  }
}

void FeatureWindow::removeFailures() {
  tracks.erase(std::remove_if(tracks.begin(), tracks.end(), [](const Track &t) { return t.solve_flag == 2; }), tracks.end());
}

void FeatureWindow::clearDepth() {
  for (Track &t : tracks) t.estimated_depth = -1.0;
}

void FeatureWindow::removeOutlier(const int *ids, int n) {
  tracks.erase(std::remove_if(tracks.begin(), tracks.end(), [&](const Track &t) { return std::find(ids, ids + n, t.feature_id) != ids + n; }),
               tracks.end());
}

void FeatureWindow::triangulate(const double *Ps, const double *Rs, const double *tic, const double *ric) {
  for (Track &t : tracks) {
    if (t.estimated_depth > 0) continue;
    double P0[12], P1[12], X[3];
    const double *p0 = t.obs[0].point, *p1 = nullptr;
    if (cfg.stereo && t.obs[0].is_stereo) {
      camera_projection(Ps, Rs, tic, ric, t.start_frame, 0, P0);
      camera_projection(Ps, Rs, tic, ric, t.start_frame, 1, P1);
      p1 = t.obs[0].point_right;
    } else if (t.obs.size() > 1) {
      camera_projection(Ps, Rs, tic, ric, t.start_frame, 0, P0);
      camera_projection(Ps, Rs, tic, ric, t.start_frame + 1, 0, P1);
      p1 = t.obs[1].point;
    } else {
      continue;   // a single mono observation: nothing to triangulate with (the reference's :380 `used_num < 4` exit)
    }
    triangulate_point(P0, P1, p0, p1, X);
    const double depth = P0[8] * X[0] + P0[9] * X[1] + P0[10] * X[2] + P0[11];
    t.estimated_depth = depth > 0 ? depth : cfg.init_depth;
  }
}

void FeatureWindow::removeBackShiftDepth(const double marg_R[9], const double marg_P[3], const double new_R[9], const double new_P[3]) {
  std::vector<Track> kept;
  kept.reserve(tracks.size());
  for (Track &t : tracks) {
    if (t.start_frame != 0) { --t.start_frame; kept.push_back(std::move(t)); continue; }
    const double uv[3] = {t.obs[0].point[0], t.obs[0].point[1], t.obs[0].point[2]};
    t.obs.erase(t.obs.begin());
    if (t.obs.size() < 2) continue;   // dropped
    double pi[3], w[3], d[3], pj[3];
    for (int k = 0; k < 3; ++k) pi[k] = uv[k] * t.estimated_depth;
    mat3_mul_vec(marg_R, pi, w);
    for (int k = 0; k < 3; ++k) d[k] = w[k] + marg_P[k] - new_P[k];
    mat3T_mul_vec(new_R, d, pj);
    t.estimated_depth = pj[2] > 0 ? pj[2] : cfg.init_depth;
    kept.push_back(std::move(t));
  }
  tracks.swap(kept);
}

void FeatureWindow::removeBack() {
  std::vector<Track> kept;
  kept.reserve(tracks.size());
  for (Track &t : tracks) {
    if (t.start_frame != 0) --t.start_frame;
    else {
      t.obs.erase(t.obs.begin());
      if (t.obs.empty()) continue;
    }
    kept.push_back(std::move(t));
  }
  tracks.swap(kept);
}

void FeatureWindow::removeFront(int frame_count) {
  std::vector<Track> kept;
  kept.reserve(tracks.size());
  for (Track &t : tracks) {
    if (t.start_frame == frame_count) --t.start_frame;
    else if (t.endFrame() >= frame_count - 1) {
      t.obs.erase(t.obs.begin() + (cfg.window_size - 1 - t.start_frame));
      if (t.obs.empty()) continue;
    }
    kept.push_back(std::move(t));
  }
  tracks.swap(kept);
}

void FeatureWindow::fill(vilo_window_desc *desc, std::vector<int32_t> *start, std::vector<int32_t> *offset, std::vector<double> *obs,
                         std::vector<uint8_t> *stereo) const {
  start->clear(); offset->assign(1, 0); obs->clear(); stereo->clear();
  for (const Track &t : tracks) {
    if (t.obs.size() < 4) continue;
    start->push_back(t.start_frame);
    for (const Observation &o : t.obs) {
      const double row[11] = {o.point[0], o.point[1], o.point[2], o.point_right[0], o.point_right[1], o.point_right[2],
                              o.velocity[0], o.velocity[1], o.velocity_right[0], o.velocity_right[1], o.cur_td};
      obs->insert(obs->end(), row, row + 11);
      stereo->push_back(o.is_stereo);
    }
    offset->push_back((int32_t)stereo->size());
  }
  desc->n_landmarks = (int32_t)start->size();
  desc->n_obs = (int32_t)stereo->size();
  desc->lm_start_frame = start->data(); desc->lm_obs_offset = offset->data(); desc->obs = obs->data(); desc->obs_is_stereo = stereo->data();
}

}  // namespace vilo

extern "C" {
using vilo::FeatureWindow;
void *vilo_fw_create() { return new FeatureWindow(); }
void vilo_fw_destroy(void *h) { delete (FeatureWindow *)h; }
int vilo_fw_add_frame(void *h, int frame_count, int n, const int *ids, const double *obs11, const uint8_t *stereo, double td, int *c3) {
  FeatureWindow *f = (FeatureWindow *)h;
  const bool kf = f->addFrame(frame_count, n, ids, obs11, stereo, td);
  if (c3) { c3[0] = f->last_track_num; c3[1] = f->new_feature_num; c3[2] = f->long_track_num; }
  return kf ? 1 : 0;
}
int vilo_fw_feature_count(void *h) { return ((FeatureWindow *)h)->featureCount(); }
void vilo_fw_depth_vector(void *h, double *out) { ((FeatureWindow *)h)->depthVector(out); }
void vilo_fw_set_depth(void *h, const double *x) { ((FeatureWindow *)h)->setDepth(x); }
void vilo_fw_remove_failures(void *h) { ((FeatureWindow *)h)->removeFailures(); }
void vilo_fw_clear_depth(void *h) { ((FeatureWindow *)h)->clearDepth(); }
void vilo_fw_remove_outlier(void *h, const int *ids, int n) { ((FeatureWindow *)h)->removeOutlier(ids, n); }
void vilo_fw_triangulate(void *h, const double *Ps, const double *Rs, const double *tic, const double *ric) { ((FeatureWindow *)h)->triangulate(Ps, Rs, tic, ric); }
void vilo_fw_remove_back_shift_depth(void *h, const double *mR, const double *mP, const double *nR, const double *nP) { ((FeatureWindow *)h)->removeBackShiftDepth(mR, mP, nR, nP); }
void vilo_fw_remove_back(void *h) { ((FeatureWindow *)h)->removeBack(); }
void vilo_fw_remove_front(void *h, int frame_count) { ((FeatureWindow *)h)->removeFront(frame_count); }
int vilo_fw_dump(void *h, int *info, double *depth, double *obs11, uint8_t *stereo, int *total_obs) {
  const FeatureWindow *f = (const FeatureWindow *)h;
  int tot = 0, k = 0;
  for (const vilo::Track &t : f->tracks) {
    if (info) { info[4 * k] = t.feature_id; info[4 * k + 1] = t.start_frame; info[4 * k + 2] = (int)t.obs.size(); info[4 * k + 3] = t.solve_flag; }
    if (depth) depth[k] = t.estimated_depth;
    for (const vilo::Observation &o : t.obs) {
      if (obs11) {
        const double row[11] = {o.point[0], o.point[1], o.point[2], o.point_right[0], o.point_right[1], o.point_right[2],
                                o.velocity[0], o.velocity[1], o.velocity_right[0], o.velocity_right[1], o.cur_td};
        std::memcpy(obs11 + 11 * tot, row, sizeof row);
      }
      if (stereo) stereo[tot] = o.is_stereo;
      ++tot;
    }
    ++k;
  }
  if (total_obs) *total_obs = tot;
  return k;
}
}
