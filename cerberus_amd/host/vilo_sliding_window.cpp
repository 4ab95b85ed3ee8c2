// See vilo_sliding_window.h. Host bookkeeping only: every factor evaluation, the trust-region solve, preintegration and the
// marginalisation run in libvilo_gpu.so.
#include "vilo_sliding_window.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include "../../include/vilo_window_io.h"
#include "../csrc/vilo_math.hpp"
#include "../csrc/worker_pool.hpp"

namespace vilo {
namespace {

// Eigen::Quaterniond(Matrix3d) (the branch structure of Eigen's quaternion-from-rotation-matrix)
quat quat_from_R(const m3 &m) {
  quat q;
  double t = m.a[0] + m.a[4] + m.a[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m.a[7] - m.a[5]) * t; q.y = (m.a[2] - m.a[6]) * t; q.z = (m.a[3] - m.a[1]) * t;
  } else {
    int i = 0;
    if (m.a[4] > m.a[0]) i = 1;
    if (m.a[8] > m.a[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m.a[4 * i] - m.a[4 * j] - m.a[4 * k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m.a[3 * k + j] - m.a[3 * j + k]) * t;
    v[j] = (m.a[3 * j + i] + m.a[3 * i + j]) * t;
    v[k] = (m.a[3 * k + i] + m.a[3 * i + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}
// Utility::R2ypr (utility.h:83-98), degrees
v3 R2ypr(const m3 &R) {
  const double y = std::atan2(R.a[3], R.a[0]);
  const double p = std::atan2(-R.a[6], R.a[0] * std::cos(y) + R.a[3] * std::sin(y));
  const double r = std::atan2(R.a[2] * std::sin(y) - R.a[5] * std::cos(y), -R.a[1] * std::sin(y) + R.a[4] * std::cos(y));
  return mk3(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);
}
// Utility::ypr2R (utility.h:100-126)
m3 ypr2R(const v3 &ypr) {
  const double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
  m3 Rz = m3_eye(), Ry = m3_eye(), Rx = m3_eye();
  Rz.a[0] = std::cos(y); Rz.a[1] = -std::sin(y); Rz.a[3] = std::sin(y); Rz.a[4] = std::cos(y);
  Ry.a[0] = std::cos(p); Ry.a[2] = std::sin(p); Ry.a[6] = -std::sin(p); Ry.a[8] = std::cos(p);
  Rx.a[4] = std::cos(r); Rx.a[5] = -std::sin(r); Rx.a[7] = std::sin(r); Rx.a[8] = std::cos(r);
  return Rz * Ry * Rx;
}
// Utility::g2R (utility.cpp:12-22) with Eigen's Quaternion::FromTwoVectors for the non-antiparallel case
m3 g2R(const v3 &gv) {
  const v3 a = gv * (1.0 / norm(gv)), b = mk3(0, 0, 1.0);
  const double c = dot(a, b);
  quat q;
  if (c < -1.0 + 1e-12) {
    q = mkq(0.0, 1.0, 0.0, 0.0);   // half turn about any axis perpendicular to gravity
  } else {
    const v3 axis = cross(a, b);
    const double s = std::sqrt((1.0 + c) * 2.0);
    q = mkq(0.5 * s, axis.x / s, axis.y / s, axis.z / s);
  }
  m3 R0 = qR(q);
  const double yaw = R2ypr(R0).x;
  return ypr2R(mk3(-yaw, 0, 0)) * R0;
}
inline void st_m3(double *p, const m3 &m) { for (int i = 0; i < 9; ++i) p[i] = m.a[i]; }
inline void cp(double *d, const double *s, int n) { std::memcpy(d, s, sizeof(double) * n); }
template <int N> inline void swp(double (&a)[N], double (&b)[N]) { for (int i = 0; i < N; ++i) std::swap(a[i], b[i]); }

}  // namespace

SlidingWindow::PriorStore::PriorStore() : x0(7 * VILO_MAX_PRIOR_BLOCKS), J0((size_t)VILO_MAX_PRIOR_DIM * VILO_MAX_PRIOR_DIM), r0(VILO_MAX_PRIOR_DIM) {
  std::memset(&p, 0, sizeof(p));
  bind();
}

SlidingWindow::SlidingWindow(vilo_ctx *ctx, const vilo_config &cfg, const SlidingWindowOptions &opt)
    : f_manager(opt.features), ctx_(ctx), cfg_(cfg), opt_(opt), pre_(NF), pre_imu_(NF) {
  resident_ = opt.resident && opt.streaming_preintegration && opt.dump_dir.empty();
  clearState();
  g[0] = 0; g[1] = 0; g[2] = cfg.g_norm;
}

SlidingWindow::~SlidingWindow() {
  if (own_pool_ && pool_) vilo_preint_streams_destroy(ctx_, pool_);
  if (own_ppool_ && ppool_) vilo_prior_pool_destroy(ctx_, ppool_);
}

void SlidingWindow::attachPriorPool(vilo_prior_pool *pool, int base_slot) {
  if (own_ppool_ && ppool_) vilo_prior_pool_destroy(ctx_, ppool_);
  ppool_ = pool; own_ppool_ = false; pslot_base_ = base_slot;
}

int SlidingWindow::priorDim() const {
  if (resident_) return ppool_ ? vilo_prior_pool_dim(ppool_, pslot_base_ + cur_prior_) : 0;
  return prior_[cur_prior_].p.valid ? prior_[cur_prior_].p.n : 0;
}

void SlidingWindow::attachStreams(vilo_preint_streams *pool, int base_id) {
  if (own_pool_ && pool_) vilo_preint_streams_destroy(ctx_, pool_);
  pool_ = pool; own_pool_ = false;
  for (int j = 0; j < NF; ++j) {
    sid_[j] = base_id + j;
    need_reset_[j] = !buf_[j].empty();
    pushed_[j] = 0;
  }
}

void SlidingWindow::clearState() {
  const m3 I = m3_eye();
  for (int i = 0; i < NF; ++i) {
    st_m3(Rs[i], I);
    for (int k = 0; k < 3; ++k) Ps[i][k] = Vs[i][k] = Bas[i][k] = Bgs[i][k] = 0.0;
    for (int k = 0; k < 4; ++k) Rho[i][k] = 0.21;   // VILO_LOWER_LEG_LENGTH (estimator.cpp:167-170, yaml lower_leg_length)
    Headers[i] = 0.0;
    buf_[i].clear();
    dirty_[i] = false;
    if (!pool_) sid_[i] = i;
    pushed_[i] = 0;
    need_reset_[i] = false;
    sum_dt_[i] = 0.0;
    std::memset(lin_[i], 0, sizeof lin_[i]);
  }
  for (int c = 0; c < 2; ++c) {
    st_m3(ric[c], I);
    tic[c][0] = tic[c][1] = tic[c][2] = 0.0;
  }
  first_imu = false; init_first_pose_flag = false;
  sum_of_back = sum_of_front = 0; frame_count = 0; solver_flag = INITIAL; open_ex_estimation = 0;
  prior_[0].p.valid = prior_[1].p.valid = 0; cur_prior_ = 0;
  if (ppool_) { vilo_prior_pool_upload(ctx_, ppool_, pslot_base_, nullptr); vilo_prior_pool_upload(ctx_, ppool_, pslot_base_ + 1, nullptr); } pending_ = 0; n_optimizations = 0;
  std::memset(&last_sample, 0, sizeof last_sample);
  std::memset(&last_summary, 0, sizeof last_summary);
  f_manager.clearState();
}

void SlidingWindow::setExtrinsics(const double *t, const double *r, double td_) {
  cp(&tic[0][0], t, 6);
  cp(&ric[0][0], r, 18);
  td = td_;
}

void SlidingWindow::initFirstPose(const double p[3], const double R[9]) {
  cp(Ps[0], p, 3);
  cp(Rs[0], R, 9);
  init_first_pose_flag = true;
}

void SlidingWindow::initFirstIMUPose(const vilo_sample *s, int n) {
  init_first_pose_flag = true;
  v3 aver = mk3(0, 0, 0);
  for (int i = 0; i < n; ++i) aver = aver + ld3(s[i].acc);
  aver = aver * (1.0 / n);
  st_m3(Rs[0], g2R(aver));
}

void SlidingWindow::startInterval(int j) {
  buf_[j].assign(1, last_sample);
  cp(lin_[j], Bas[j], 3); cp(lin_[j] + 3, Bgs[j], 3); cp(lin_[j] + 6, Rho[j], 4);
  dirty_[j] = true;
  need_reset_[j] = true; pushed_[j] = 0;
  sum_dt_[j] = 0.0;
}

void SlidingWindow::processIMULeg(const vilo_sample &s) {
  if (!first_imu) {
    first_imu = true;
    last_sample = s;
  }
  if (buf_[frame_count].empty()) startInterval(frame_count);
  if (frame_count != 0) {
    const int j = frame_count;
    buf_[j].push_back(s);
    dirty_[j] = true;
    sum_dt_[j] += s.dt;
    // mid-point propagation of the newest frame (estimator.cpp:634-641)
    const double dt = s.dt;
    const v3 gv = ld3(g), ba = ld3(Bas[j]), bg = ld3(Bgs[j]);
    m3 R = ld_m3_rowmajor(Rs[j]);
    const v3 un_acc_0 = R * (ld3(last_sample.acc) - ba) - gv;
    const v3 un_gyr = (ld3(last_sample.gyr) + ld3(s.gyr)) * 0.5 - bg;
    R = R * qR(deltaQ(un_gyr * dt));
    const v3 un_acc_1 = R * (ld3(s.acc) - ba) - gv;
    const v3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
    const v3 P = ld3(Ps[j]) + ld3(Vs[j]) * dt + un_acc * (0.5 * dt * dt);
    const v3 V = ld3(Vs[j]) + un_acc * dt;
    st_m3(Rs[j], R); st3(Ps[j], P); st3(Vs[j], V);
  }
  last_sample = s;
}

void SlidingWindow::vector2double() {
  para_Pose.resize(7 * NF); para_SpeedBias.resize(9 * NF); para_LegBias.resize(4 * NF); para_Ex_Pose.resize(14); para_Td.resize(1);
  for (int i = 0; i < NF; ++i) {
    double *p = &para_Pose[7 * i];
    cp(p, Ps[i], 3);
    const quat q = quat_from_R(ld_m3_rowmajor(Rs[i]));
    p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
    double *sb = &para_SpeedBias[9 * i];
    cp(sb, Vs[i], 3); cp(sb + 3, Bas[i], 3); cp(sb + 6, Bgs[i], 3);
    cp(&para_LegBias[4 * i], Rho[i], 4);
  }
  for (int c = 0; c < 2; ++c) {
    double *p = &para_Ex_Pose[7 * c];
    cp(p, tic[c], 3);
    const quat q = quat_from_R(ld_m3_rowmajor(ric[c]));
    p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
  }
  para_Feature.assign(std::max(1, f_manager.featureCount()), 0.0);
  f_manager.depthVector(para_Feature.data());
  para_Td[0] = td;
}

void SlidingWindow::double2vector() {
  // the yaw / position re-anchoring itself (estimator.cpp:905-957) ran in vilo_gauge_fix on para_*; unpack
  for (int i = 0; i < NF; ++i) {
    const double *p = &para_Pose[7 * i];
    cp(Ps[i], p, 3);
    st_m3(Rs[i], qR(qnormalized(ldq_pose(p))));
    const double *sb = &para_SpeedBias[9 * i];
    cp(Vs[i], sb, 3); cp(Bas[i], sb + 3, 3); cp(Bgs[i], sb + 6, 3);
    if (opt_.use_leg) cp(Rho[i], &para_LegBias[4 * i], 4);
  }
  for (int c = 0; c < 2; ++c) {
    const double *p = &para_Ex_Pose[7 * c];
    cp(tic[c], p, 3);
    st_m3(ric[c], qR(qnormalized(ldq_pose(p))));
  }
  if (f_manager.featureCount() > 0) f_manager.setDepth(para_Feature.data());
  td = para_Td[0];
}

void SlidingWindow::fillDesc() {
  std::memset(&desc_, 0, sizeof desc_);
  f_manager.fill(&desc_, &lm_start_, &lm_off_, &obs_, &stereo_);
  // para_Feature has NUM_OF_F rows (estimator.h:196, parameters.h:24): the reference relies on the tracker's MAX_CNT to stay below
  // it; here the features beyond the first NUM_OF_F of the list are left out of the problem (their depths stay as triangulated)
  if (desc_.n_landmarks > VILO_NUM_OF_F) {
    desc_.n_landmarks = VILO_NUM_OF_F;
    desc_.n_obs = lm_off_[VILO_NUM_OF_F];
  }
  desc_.n_frames = frame_count + 1;
  desc_.use_leg = opt_.use_leg;
  desc_.preint = opt_.use_leg ? &pre_[1] : nullptr;
  desc_.preint_imu = opt_.use_leg ? nullptr : &pre_imu_[1];
  desc_.prior = (!resident_ && hasPrior()) ? &prior_[cur_prior_].p : nullptr;
  if (resident_) { desc_.preint = nullptr; desc_.preint_imu = nullptr; }
  // SetParameterBlockConstant decisions, estimator.cpp:1074-1106
  desc_.leg_bias_const = (opt_.use_leg && !opt_.optimize_leg_bias) || frame_count < WS;
  const double v0 = norm(ld3(Vs[0]));
  if ((opt_.estimate_extrinsic && frame_count == WS && v0 > 0.2) || open_ex_estimation) {
    open_ex_estimation = 1;
    desc_.ex_const = 0;
  } else {
    desc_.ex_const = 1;
  }
  desc_.td_const = (!opt_.estimate_td || v0 < 0.2) ? 1 : 0;
  state_.pose = para_Pose.data(); state_.speed_bias = para_SpeedBias.data(); state_.leg_bias = para_LegBias.data();
  state_.ex_pose = para_Ex_Pose.data(); state_.td = para_Td.data(); state_.inv_depth = para_Feature.data();
}

bool SlidingWindow::beginImage(double header, int n, const int *ids, const double *obs11, const uint8_t *stereo) {
  marginalization_flag = f_manager.addFrame(frame_count, n, ids, obs11, stereo, td) ? MARGIN_OLD : MARGIN_SECOND_NEW;
  Headers[frame_count] = header;
  pending_ = 0;
  if (solver_flag == INITIAL) {
    f_manager.triangulate(&Ps[0][0], &Rs[0][0], &tic[0][0], &ric[0][0]);
    if (frame_count == WS) {
      // il_pre_integrations[i]->repropagate(Zero, Bgs[i], rho_i) (estimator.cpp:752-760)
      for (int i = 0; i < NF; ++i) {
        lin_[i][0] = lin_[i][1] = lin_[i][2] = 0.0;
        cp(lin_[i] + 3, Bgs[i], 3); cp(lin_[i] + 6, Rho[i], 4);
        dirty_[i] = true;
        need_reset_[i] = true; pushed_[i] = 0;   // repropagate = constructor + every buffered push_back again
        sum_dt_[i] = 0.0;
        for (size_t q = 1; q < buf_[i].size(); ++q) sum_dt_[i] += buf_[i][q].dt;
      }
      pending_ = 1;
      return true;
    }
    // estimator.cpp:789-802
    ++frame_count;
    const int prev = frame_count - 1;
    cp(Ps[frame_count], Ps[prev], 3); cp(Vs[frame_count], Vs[prev], 3); cp(Rs[frame_count], Rs[prev], 9);
    cp(Bas[frame_count], Bas[prev], 3); cp(Bgs[frame_count], Bgs[prev], 3); cp(Rho[frame_count], Rho[prev], 4);
    return false;
  }
  f_manager.triangulate(&Ps[0][0], &Rs[0][0], &tic[0][0], &ric[0][0]);
  pending_ = 2;
  return true;
}

void SlidingWindow::endImage() {
  if (pending_ == 1) {
    solver_flag = NON_LINEAR;
    slideWindow();
  } else if (pending_ == 2) {
    std::vector<int> remove_ids;
    outliersRejection(&remove_ids);
    f_manager.removeOutlier(remove_ids.data(), (int)remove_ids.size());
    slideWindow();
    f_manager.removeFailures();
  }
  pending_ = 0;
}

int SlidingWindow::processImage(double header, int n, const int *ids, const double *obs11, const uint8_t *stereo) {
  int rc = VILO_OK;
  if (beginImage(header, n, ids, obs11, stereo)) {
    SlidingWindow *self = this;
    rc = optimizeBatch(ctx_, &self, 1);
  }
  // VILO_ERR_NUMERIC is a per-window condition (a failed solve leaves the states alone, a non-finite marginalisation drops the prior): the
  // window still slides, so that it stays usable; the code is passed on to the caller
  if (rc == VILO_OK || rc == VILO_ERR_NUMERIC) endImage();
  return rc;
}

int SlidingWindow::dump(const vilo_window_state &before) const {
  char path[1024];
  std::snprintf(path, sizeof path, "%s/win_%05d.bin", opt_.dump_dir.c_str(), n_optimizations);
  const double ref_summary[4] = {(double)last_summary.iterations, last_summary.initial_cost, last_summary.final_cost, (double)last_summary.termination};
  return vilo_window_write(path, &cfg_, &desc_, &before, &state_, ref_summary, marginalization_flag);
}

double now_ms_public();
void parallel_for_public(int n, const std::function<void(int)> &fn);
namespace {
// robots are independent: host bookkeeping of a fleet runs on a few threads (worker_pool.hpp: started once, parked between calls —
// an image step of a fleet makes four of these calls)
void parallel_for(int n, const std::function<void(int)> &fn) { parallel_items(n, 4, fn); }
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

// The device-resident preintegration objects brought up to date with the windows' sample buffers: the objects of intervals whose
// linearisation point changed are reset (constructor), every sample not pushed yet is push_back()ed, one reset and one push launch per
// pool for all the windows that share it. read_back (the image step of a window whose prior is carried on the host): the records of the
// changed intervals come back into pre_ / pre_imu_. Without it this is what a caller runs BETWEEN images (pushSamples): the reference
// integrates every IMU / leg message as it arrives (processIMULeg -> push_back, estimator.cpp:612-632), and pushing an interval in pieces
// gives bitwise the record of pushing it at once, so how often it is called changes when the work is done, not the result.
int SlidingWindow::pushPending(vilo_ctx *ctx, SlidingWindow *const *ws, int n, bool read_back) {
  const int use_leg = ws[0]->opt_.use_leg;
  const bool resident = ws[0]->resident_;
  for (int w0 = 0; w0 < n;) {
    int w1 = w0 + 1;
    while (w1 < n && ws[w1]->pool_ == ws[w0]->pool_) ++w1;
    vilo_preint_streams *pool = ws[w0]->pool_;
    std::vector<int32_t> rid, pid, gid, offsets(1, 0);
    std::vector<vilo_sample> first, samples;
    std::vector<double> lin;
    std::vector<std::pair<int, int>> which;
    for (int w = w0; w < w1; ++w) {
      SlidingWindow &s = *ws[w];
      for (int j = 1; j <= WS; ++j) {
        if (s.buf_[j].empty()) continue;
        if (s.need_reset_[j]) {
          rid.push_back(s.sid_[j]); first.push_back(s.buf_[j][0]); lin.insert(lin.end(), s.lin_[j], s.lin_[j] + (use_leg ? 10 : 6));
          s.need_reset_[j] = false; s.pushed_[j] = 0;
        }
        const int have = (int)s.buf_[j].size() - 1;
        if (s.pushed_[j] < have) {
          pid.push_back(s.sid_[j]);
          samples.insert(samples.end(), s.buf_[j].begin() + 1 + s.pushed_[j], s.buf_[j].end());
          offsets.push_back((int32_t)samples.size());
          s.pushed_[j] = have;
        }
        if (!read_back) continue;
        if (s.dirty_[j] && !resident) { gid.push_back(s.sid_[j]); which.push_back({w, j}); }
        if (resident) s.dirty_[j] = false;
      }
    }
    int rc = vilo_preint_streams_reset(ctx, pool, (int)rid.size(), rid.data(), first.data(), lin.data());
    if (rc == VILO_OK) rc = vilo_preint_streams_push(ctx, pool, (int)pid.size(), pid.data(), samples.data(), offsets.data());
    if (rc != VILO_OK) return rc;
    if (read_back) {
      if (use_leg) {
        std::vector<vilo_preint> out(gid.size());
        rc = vilo_preint_streams_read(ctx, pool, (int)gid.size(), gid.data(), out.data());
        if (rc != VILO_OK) return rc;
        for (size_t k = 0; k < which.size(); ++k) ws[which[k].first]->pre_[which[k].second] = out[k];
      } else {
        std::vector<vilo_preint_imu> out(gid.size());
        rc = vilo_preint_streams_read_imu(ctx, pool, (int)gid.size(), gid.data(), out.data());
        if (rc != VILO_OK) return rc;
        for (size_t k = 0; k < which.size(); ++k) ws[which[k].first]->pre_imu_[which[k].second] = out[k];
      }
      for (size_t k = 0; k < which.size(); ++k) ws[which[k].first]->dirty_[which[k].second] = false;
    }
    w0 = w1;
  }
  return VILO_OK;
}

// What the reference does per message, for a fleet between two images: the samples buffered since the last call go to the windows'
// device-resident preintegration objects now (windows without streaming preintegration, or with nothing new, are left alone).
int SlidingWindow::pushSamples(vilo_ctx *ctx, SlidingWindow *const *ws, int n) {
  std::vector<SlidingWindow *> act;
  for (int w = 0; w < n; ++w) {
    SlidingWindow &s = *ws[w];
    if (!s.opt_.streaming_preintegration) continue;
    if (!s.pool_) {
      int rc = s.opt_.use_leg ? vilo_preint_streams_create(ctx, NF, &s.pool_) : vilo_preint_streams_create_imu(ctx, NF, &s.pool_);
      if (rc != VILO_OK) return rc;
      s.own_pool_ = true;
      for (int j = 0; j < NF; ++j) { s.sid_[j] = j; s.need_reset_[j] = !s.buf_[j].empty(); s.pushed_[j] = 0; }
    }
    act.push_back(&s);
  }
  // one call of pushPending serves windows of one factor kind; windows sharing a pool are of one kind and are kept adjacent
  std::stable_sort(act.begin(), act.end(), [](const SlidingWindow *a, const SlidingWindow *b) {
    return a->opt_.use_leg != b->opt_.use_leg ? a->opt_.use_leg < b->opt_.use_leg : std::less<const void *>()(a->pool_, b->pool_);
  });
  for (size_t i = 0; i < act.size();) {
    size_t k = i + 1;
    while (k < act.size() && act[k]->opt_.use_leg == act[i]->opt_.use_leg) ++k;
    const int rc = pushPending(ctx, act.data() + i, (int)(k - i), /*read_back=*/false);
    if (rc != VILO_OK) return rc;
    i = k;
  }
  return VILO_OK;
}

int SlidingWindow::optimizeBatch(vilo_ctx *ctx, SlidingWindow *const *ws, int n) {
  if (n <= 0) return VILO_OK;
  const bool timing = getenv("VILO_HOST_TIMING") != nullptr;
  const double t0 = now_ms();
  const int use_leg = ws[0]->opt_.use_leg;
  for (int w = 0; w < n; ++w)
    if (ws[w]->opt_.use_leg != use_leg || ws[w]->pending_ == 0 || ws[w]->frame_count != WS) return VILO_ERR_BAD_ARG;
  // 1. preintegration of the intervals whose samples or linearisation point changed: one device call sequence for the whole fleet
  bool streaming = ws[0]->opt_.streaming_preintegration != 0;
  // robots without pools get their own; a fleet shares pools (then every robot is served by the same few device calls)
  for (int w = 0; w < n; ++w) {
    SlidingWindow &s = *ws[w];
    if (s.opt_.streaming_preintegration && !s.pool_) {
      int rc = s.opt_.use_leg ? vilo_preint_streams_create(ctx, NF, &s.pool_) : vilo_preint_streams_create_imu(ctx, NF, &s.pool_);
      if (rc != VILO_OK) return rc;
      s.own_pool_ = true;
      for (int j = 0; j < NF; ++j) { s.sid_[j] = j; s.need_reset_[j] = !s.buf_[j].empty(); s.pushed_[j] = 0; }
    }
    if (s.resident_ && !s.ppool_) {
      int rc = vilo_prior_pool_create(ctx, 2, &s.ppool_);
      if (rc != VILO_OK) return rc;
      s.own_ppool_ = true; s.pslot_base_ = 0;
    }
  }
  // one call serves windows of one kind only (same options, same pools): split a mixed fleet
  auto same_kind = [](const SlidingWindow &a, const SlidingWindow &b) {
    return a.resident_ == b.resident_ && a.pool_ == b.pool_ && a.ppool_ == b.ppool_ && a.opt_.streaming_preintegration == b.opt_.streaming_preintegration;
  };
  int soft_rc = VILO_OK;
  for (int w = 1; w < n; ++w)
    if (!same_kind(*ws[w], *ws[0])) {
      std::vector<SlidingWindow *> rest(ws, ws + n), grp;
      while (!rest.empty()) {
        grp.clear();
        std::vector<SlidingWindow *> other;
        for (SlidingWindow *x : rest) (same_kind(*x, *rest[0]) ? grp : other).push_back(x);
        const int rc = optimizeBatch(ctx, grp.data(), (int)grp.size());
        if (rc != VILO_OK && rc != VILO_ERR_NUMERIC) return rc;
        if (rc == VILO_ERR_NUMERIC) soft_rc = rc;
        rest.swap(other);
      }
      return soft_rc;
    }
  const bool resident = ws[0]->resident_;
  if (streaming) {
    const int rc = pushPending(ctx, ws, n, /*read_back=*/true);
    if (rc != VILO_OK) return rc;
  } else {
    std::vector<vilo_sample> samples;
    std::vector<int32_t> offsets(1, 0);
    std::vector<double> lin;
    std::vector<std::pair<int, int>> which;
    for (int w = 0; w < n; ++w)
      for (int j = 1; j <= WS; ++j) {
        SlidingWindow &s = *ws[w];
        if (!s.dirty_[j]) continue;
        samples.insert(samples.end(), s.buf_[j].begin(), s.buf_[j].end());
        offsets.push_back((int32_t)samples.size());
        if (use_leg) lin.insert(lin.end(), s.lin_[j], s.lin_[j] + 10);
        else lin.insert(lin.end(), s.lin_[j], s.lin_[j] + 6);
        which.push_back({w, j});
      }
    if (!which.empty()) {
      int rc;
      if (use_leg) {
        std::vector<vilo_preint> out(which.size());
        rc = vilo_preintegrate(ctx, (int)which.size(), samples.data(), offsets.data(), lin.data(), out.data());
        if (rc != VILO_OK) return rc;
        for (size_t k = 0; k < which.size(); ++k) ws[which[k].first]->pre_[which[k].second] = out[k];
      } else {
        std::vector<vilo_preint_imu> out(which.size());
        rc = vilo_preintegrate_imu(ctx, (int)which.size(), samples.data(), offsets.data(), lin.data(), out.data());
        if (rc != VILO_OK) return rc;
        for (size_t k = 0; k < which.size(); ++k) ws[which[k].first]->pre_imu_[which[k].second] = out[k];
      }
      for (auto &wj : which) ws[wj.first]->dirty_[wj.second] = false;
    }
  }
  const double t1 = now_ms();
  // 2. vector2double + the problem description
  std::vector<vilo_window_desc> descs(n);
  std::vector<vilo_window_state> states(n), befores(n);
  std::vector<std::vector<double>> keep(n);   // pre-solve copies (the dump's `before`)
  std::vector<vilo_solve_summary> sums(n);
  parallel_for(n, [&](int w) {
    SlidingWindow &s = *ws[w];
    s.vector2double();
    s.fillDesc();
    descs[w] = s.desc_; states[w] = s.state_;
    std::vector<double> &k = keep[w];
    k.insert(k.end(), s.para_Pose.begin(), s.para_Pose.end());
    k.insert(k.end(), s.para_SpeedBias.begin(), s.para_SpeedBias.end());
    k.insert(k.end(), s.para_LegBias.begin(), s.para_LegBias.end());
    k.insert(k.end(), s.para_Ex_Pose.begin(), s.para_Ex_Pose.end());
    k.insert(k.end(), s.para_Td.begin(), s.para_Td.end());
    k.insert(k.end(), s.para_Feature.begin(), s.para_Feature.end());
    double *b = k.data();
    befores[w].pose = b; b += 7 * NF;
    befores[w].speed_bias = b; b += 9 * NF;
    befores[w].leg_bias = b; b += 4 * NF;
    befores[w].ex_pose = b; b += 14;
    befores[w].td = b; b += 1;
    befores[w].inv_depth = b;
  });
  // 3. Estimator::optimization() from ceres::Solve on (estimator.cpp:1236-1455) in one device call: solve, double2vector's gauge
  //    fix, marginalisation at the result — one packing of the batch, no host round trip between the halves
  const double t2 = now_ms();
  std::vector<int> flags(n);
  std::vector<vilo_prior> next(n);
  std::vector<vilo_resident_refs> refs(resident ? n : 0);
  for (int w = 0; w < n; ++w) {
    SlidingWindow &s = *ws[w];
    flags[w] = s.marginalization_flag;
    if (resident) {
      vilo_resident_refs &r = refs[w];
      r.preint_pool = s.pool_; r.preint_ids = &s.sid_[1]; r.preint_sum_dt = &s.sum_dt_[1];
      r.prior_pool = s.ppool_; r.prior_slot = s.pslot_base_ + s.cur_prior_; r.next_prior_slot = s.pslot_base_ + 1 - s.cur_prior_;
      continue;
    }
    PriorStore &nx = s.prior_[1 - s.cur_prior_];
    nx.bind();
    next[w] = nx.p;
  }
  int rc = resident ? vilo_optimize_windows_resident(ctx, n, descs.data(), refs.data(), states.data(), &ws[0]->opt_.solve, flags.data(), nullptr, sums.data())
                    : vilo_optimize_windows(ctx, n, descs.data(), states.data(), &ws[0]->opt_.solve, flags.data(), next.data(), sums.data());
  // VILO_ERR_NUMERIC: some window's solve failed (its states are untouched) or its marginalisation was not finite (its prior is invalid);
  // every window's outputs are filled in, so the bookkeeping below runs for the whole fleet and the code is returned at the end
  if (rc != VILO_OK && rc != VILO_ERR_NUMERIC) return rc;
  const double t3 = now_ms();
  std::vector<int> dump_rc(n, 0);
  parallel_for(n, [&](int w) {
    SlidingWindow &s = *ws[w];
    s.last_summary = sums[w];
    if (!s.opt_.dump_dir.empty()) dump_rc[w] = s.dump(befores[w]);
    s.double2vector();
    if (!resident) {
      s.prior_[1 - s.cur_prior_].p = next[w];
      s.prior_[1 - s.cur_prior_].bind();
    }
    s.cur_prior_ = 1 - s.cur_prior_;
    ++s.n_optimizations;
  });
  for (int w = 0; w < n; ++w)
    if (dump_rc[w] != 0) return VILO_ERR_BAD_ARG;
  if (timing)
    fprintf(stderr, "[optimizeBatch] n=%d preintegrate %.2f ms, vector2double+tables %.2f ms, vilo_optimize_windows %.2f ms, double2vector %.2f ms\n", n, t1 - t0,
            t2 - t1, t3 - t2, now_ms() - t3);
  return rc;
}

void SlidingWindow::outliersRejection(std::vector<int> *remove_ids) const {
  const m3 r0 = ld_m3_rowmajor(ric[0]), r1 = ld_m3_rowmajor(ric[1]);
  const v3 t0 = ld3(tic[0]), t1 = ld3(tic[1]);
  auto reproj = [&](int i, int j, const m3 &ricj, const v3 &ticj, double depth, const v3 &uvi, const double *uvj) {
    const m3 Ri = ld_m3_rowmajor(Rs[i]), Rj = ld_m3_rowmajor(Rs[j]);
    const v3 pts_w = Ri * (r0 * (uvi * depth) + t0) + ld3(Ps[i]);
    const v3 pc = tr(ricj) * (tr(Rj) * (pts_w - ld3(Ps[j])) - ticj);
    const double rx = pc.x / pc.z - uvj[0], ry = pc.y / pc.z - uvj[1];
    return std::sqrt(rx * rx + ry * ry);
  };
  for (const Track &t : f_manager.tracks) {
    if (t.obs.size() < 4) continue;
    double err = 0;
    int cnt = 0;
    const int imu_i = t.start_frame;
    int imu_j = imu_i - 1;
    const v3 pts_i = ld3(t.obs[0].point);
    const double depth = t.estimated_depth;
    for (const Observation &o : t.obs) {
      ++imu_j;
      if (imu_i != imu_j) { err += reproj(imu_i, imu_j, r0, t0, depth, pts_i, o.point); ++cnt; }
      if (f_manager.cfg.stereo && o.is_stereo) { err += reproj(imu_i, imu_j, r1, t1, depth, pts_i, o.point_right); ++cnt; }
    }
    if (err / cnt * f_manager.cfg.focal_length > 3) remove_ids->push_back(t.feature_id);
  }
}

void SlidingWindow::slideWindow() {
  if (frame_count != WS) return;
  if (marginalization_flag == MARGIN_OLD) {
    cp(back_R0_, Rs[0], 9); cp(back_P0_, Ps[0], 3);
    for (int i = 0; i < WS; ++i) {
      std::swap(Headers[i], Headers[i + 1]);
      swp(Rs[i], Rs[i + 1]); swp(Ps[i], Ps[i + 1]); swp(Vs[i], Vs[i + 1]); swp(Bas[i], Bas[i + 1]); swp(Bgs[i], Bgs[i + 1]);
      swp(Rho[i], Rho[i + 1]);
      buf_[i].swap(buf_[i + 1]);
      swp(lin_[i], lin_[i + 1]);
      std::swap(dirty_[i], dirty_[i + 1]);
      std::swap(sid_[i], sid_[i + 1]); std::swap(pushed_[i], pushed_[i + 1]); std::swap(need_reset_[i], need_reset_[i + 1]);
      std::swap(sum_dt_[i], sum_dt_[i + 1]);
    }
    // il_pre_integrations / pre_integrations: the same pointer swap chain as one rotation of the record arrays
    if (opt_.use_leg) std::rotate(pre_.begin(), pre_.begin() + 1, pre_.end());
    else std::rotate(pre_imu_.begin(), pre_imu_.begin() + 1, pre_imu_.end());
    Headers[WS] = Headers[WS - 1];
    cp(Ps[WS], Ps[WS - 1], 3); cp(Rs[WS], Rs[WS - 1], 9); cp(Vs[WS], Vs[WS - 1], 3); cp(Bas[WS], Bas[WS - 1], 3);
    cp(Bgs[WS], Bgs[WS - 1], 3); cp(Rho[WS], Rho[WS - 1], 4);
    startInterval(WS);
    slideWindowOld();
  } else {
    Headers[WS - 1] = Headers[WS];
    cp(Ps[WS - 1], Ps[WS], 3); cp(Rs[WS - 1], Rs[WS], 9);
    // the newest interval is appended to the one before it (estimator.cpp:1581-1599); its constructor sample is not a push
    buf_[WS - 1].insert(buf_[WS - 1].end(), buf_[WS].begin() + 1, buf_[WS].end());
    for (size_t q = 1; q < buf_[WS].size(); ++q) sum_dt_[WS - 1] += buf_[WS][q].dt;
    dirty_[WS - 1] = true;
    cp(Vs[WS - 1], Vs[WS], 3); cp(Bas[WS - 1], Bas[WS], 3); cp(Bgs[WS - 1], Bgs[WS], 3); cp(Rho[WS - 1], Rho[WS], 4);
    startInterval(WS);
    slideWindowNew();
  }
}

void SlidingWindow::slideWindowNew() {
  ++sum_of_front;
  f_manager.removeFront(frame_count);
}

void SlidingWindow::slideWindowOld() {
  ++sum_of_back;
  if (solver_flag == NON_LINEAR) {
    const m3 bR = ld_m3_rowmajor(back_R0_), R0n = ld_m3_rowmajor(Rs[0]), r0 = ld_m3_rowmajor(ric[0]);
    double mR[9], nR[9], mP[3], nP[3];
    st_m3(mR, bR * r0); st_m3(nR, R0n * r0);
    st3(mP, ld3(back_P0_) + bR * ld3(tic[0]));
    st3(nP, ld3(Ps[0]) + R0n * ld3(tic[0]));
    f_manager.removeBackShiftDepth(mR, mP, nR, nP);
  } else {
    f_manager.removeBack();
  }
}

double now_ms_public() { return now_ms(); }
void parallel_for_public(int n, const std::function<void(int)> &fn) { parallel_for(n, fn); }

}  // namespace vilo

extern "C" {
using vilo::SlidingWindow;
void *vilo_sw_create(vilo_ctx *ctx, const vilo_config *cfg, const vilo_sw_options *o) {
  vilo::SlidingWindowOptions opt;
  if (o) {
    opt.use_leg = o->use_leg; opt.optimize_leg_bias = o->optimize_leg_bias; opt.estimate_extrinsic = o->estimate_extrinsic;
    opt.estimate_td = o->estimate_td;
    if (o->max_num_iterations > 0) opt.solve.max_num_iterations = o->max_num_iterations;
    opt.solve.fixed_iterations = o->fixed_iterations;
    if (o->dump_dir) opt.dump_dir = o->dump_dir;
    opt.streaming_preintegration = o->streaming_preintegration;
    opt.resident = o->resident;
  }
  opt.features.focal_length = cfg->focal_length;
  return new SlidingWindow(ctx, *cfg, opt);
}
void vilo_sw_destroy(void *h) { delete (SlidingWindow *)h; }
void vilo_sw_attach_streams(void *h, vilo_preint_streams *pool, int base_id) { ((SlidingWindow *)h)->attachStreams(pool, base_id); }
void vilo_sw_attach_prior_pool(void *h, vilo_prior_pool *pool, int base_slot) { ((SlidingWindow *)h)->attachPriorPool(pool, base_slot); }
void vilo_sw_set_extrinsics(void *h, const double *t, const double *r, double td) { ((SlidingWindow *)h)->setExtrinsics(t, r, td); }
void vilo_sw_init_first_pose(void *h, const double *p, const double *R, const double *v) {
  SlidingWindow *s = (SlidingWindow *)h;
  s->initFirstPose(p, R);
  if (v) s->setInitialVelocity(v);
}
void vilo_sw_init_first_imu_pose(void *h, const vilo_sample *s, int n) { ((SlidingWindow *)h)->initFirstIMUPose(s, n); }
void vilo_sw_process_samples(void *h, const vilo_sample *s, int n) {
  for (int i = 0; i < n; ++i) ((SlidingWindow *)h)->processIMULeg(s[i]);
}
// self-check of the fleet's worker pool (tests/test_sliding_window.py): `repeats` jobs of n items each; 0 if every item ran exactly once
// per job and no two jobs overlapped
int vilo_sw_parallel_selfcheck(int n, int repeats) {
  std::vector<std::atomic<int>> hits((size_t)std::max(n, 1));
  for (auto &h : hits) h.store(0);
  std::atomic<int> bad{0};
  for (int r = 0; r < repeats; ++r) {
    vilo::parallel_for_public(n, [&](int i) {
      if (i < 0 || i >= n || hits[(size_t)i].fetch_add(1) != r) bad.fetch_add(1);
    });
    for (int i = 0; i < n; ++i)
      if (hits[(size_t)i].load() != r + 1) bad.fetch_add(1);
  }
  return bad.load();
}
int vilo_sw_push_samples(vilo_ctx *ctx, void *const *hs, int n) { return SlidingWindow::pushSamples(ctx, (SlidingWindow *const *)hs, n); }
int vilo_sw_process_image(void *h, double header, int n, const int *ids, const double *obs11, const uint8_t *stereo) {
  return ((SlidingWindow *)h)->processImage(header, n, ids, obs11, stereo);
}
int vilo_sw_process_images(vilo_ctx *ctx, void *const *hs, int W, const double *headers, const int *off, const int *ids, const double *obs11,
                           const uint8_t *stereo) {
  std::vector<SlidingWindow *> due;
  const double t0 = vilo::now_ms_public();
  std::vector<char> is_due(W, 0);
  vilo::parallel_for_public(W, [&](int w) {
    SlidingWindow *s = (SlidingWindow *)hs[w];
    is_due[w] = s->beginImage(headers[w], off[w + 1] - off[w], ids + off[w], obs11 + 11 * (size_t)off[w], stereo + off[w]) ? 1 : 0;
  });
  for (int w = 0; w < W; ++w)
    if (is_due[w]) due.push_back((SlidingWindow *)hs[w]);
  const double t1 = vilo::now_ms_public();
  const int rc = SlidingWindow::optimizeBatch(ctx, due.data(), (int)due.size());
  if (rc != VILO_OK && rc != VILO_ERR_NUMERIC) return rc;
  const double t2 = vilo::now_ms_public();
  vilo::parallel_for_public(W, [&](int w) { ((SlidingWindow *)hs[w])->endImage(); });
  if (getenv("VILO_HOST_TIMING"))
    fprintf(stderr, "[vilo_sw_process_images] W=%d beginImage %.2f ms, optimizeBatch %.2f ms, endImage %.2f ms\n", W, t1 - t0, t2 - t1, vilo::now_ms_public() - t2);
  return rc;
}
void vilo_sw_get_state(void *h, int *flags, double *Ps, double *Rs, double *Vs, double *Bas, double *Bgs, double *Rho, double *tic, double *ric,
                       double *td) {
  const SlidingWindow *s = (const SlidingWindow *)h;
  if (flags) {
    flags[0] = s->frame_count; flags[1] = s->solver_flag; flags[2] = s->marginalization_flag; flags[3] = s->n_optimizations;
    flags[4] = s->f_manager.featureCount(); flags[5] = s->priorDim();
  }
  const int NF = SlidingWindow::NF;
  if (Ps) std::memcpy(Ps, s->Ps, sizeof(double) * 3 * NF);
  if (Rs) std::memcpy(Rs, s->Rs, sizeof(double) * 9 * NF);
  if (Vs) std::memcpy(Vs, s->Vs, sizeof(double) * 3 * NF);
  if (Bas) std::memcpy(Bas, s->Bas, sizeof(double) * 3 * NF);
  if (Bgs) std::memcpy(Bgs, s->Bgs, sizeof(double) * 3 * NF);
  if (Rho) std::memcpy(Rho, s->Rho, sizeof(double) * 4 * NF);
  if (tic) std::memcpy(tic, s->tic, sizeof(double) * 6);
  if (ric) std::memcpy(ric, s->ric, sizeof(double) * 18);
  if (td) *td = s->td;
}
int vilo_sw_last_summary(void *h, vilo_solve_summary *out) {
  *out = ((SlidingWindow *)h)->last_summary;
  return 0;
}
}
