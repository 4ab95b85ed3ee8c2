// See vilo_sensor_buffer.h.
#include "vilo_sensor_buffer.h"
#include <chrono>

#include <cstring>

namespace vilo {

void SensorBuffer::inputIMU(double t, const double acc[3], const double gyr[3]) {
  ImuMsg m;
  m.t = t;
  std::memcpy(m.acc, acc, sizeof m.acc); std::memcpy(m.gyr, gyr, sizeof m.gyr);
  imu_.push_back(m);
}

void SensorBuffer::inputLeg(double t, const double phi[12], const double dphi[12], const double c[4]) {
  LegMsg m;
  m.t = t;
  std::memcpy(m.phi, phi, sizeof m.phi); std::memcpy(m.dphi, dphi, sizeof m.dphi); std::memcpy(m.c, c, sizeof m.c);
  leg_.push_back(m);
}

bool SensorBuffer::getIMUAndLegInterval(double t0, double t1, std::vector<ImuMsg> *imu, std::vector<LegMsg> *leg) {
  if (imu_.empty()) return false;                       // "not receive imu nor leg"
  if (!(t1 <= imu_.back().t)) return false;             // "wait for imu and leg"
  // the reference pops the five queues in lockstep; a leg queue that runs dry is undefined behaviour there — here the interval
  // is refused instead
  while (!imu_.empty() && imu_.front().t <= t0) {
    imu_.pop_front();
    if (leg_.empty()) return false;
    leg_.pop_front();
  }
  while (!imu_.empty() && imu_.front().t < t1) {
    if (leg_.empty()) return false;
    imu->push_back(imu_.front()); imu_.pop_front();
    leg->push_back(leg_.front()); leg_.pop_front();
  }
  if (imu_.empty() || leg_.empty()) return false;
  imu->push_back(imu_.front());
  leg->push_back(leg_.front());
  return true;
}

int MeasurementProcessor::inputFeature(const FeatureFrame &f) {
  featureBuf.push_back(f);
  return processMeasurements();
}

int MeasurementProcessor::processMeasurements() {
  int done = 0;
  while (!featureBuf.empty()) {
    const FeatureFrame &feature = featureBuf.front();
    curTime = feature.t + est->td;
    if (!buf.IMUAvailable(feature.t + est->td)) return done;   // "wait for imu and leg ..." (single-threaded: return)
    std::vector<ImuMsg> imu;
    std::vector<LegMsg> leg;
    if (!buf.getIMUAndLegInterval(prevTime, curTime, &imu, &leg)) return done;
    // dt of each message (:456-462): the first runs from the previous image, the last one up to this image
    last_interval.assign(imu.size(), vilo_sample());
    for (size_t i = 0; i < imu.size(); ++i) {
      vilo_sample &s = last_interval[i];
      if (i == 0) s.dt = imu[i].t - prevTime;
      else if (i + 1 == imu.size()) s.dt = curTime - imu[i - 1].t;
      else s.dt = imu[i].t - imu[i - 1].t;
      std::memcpy(s.acc, imu[i].acc, sizeof s.acc); std::memcpy(s.gyr, imu[i].gyr, sizeof s.gyr);
      std::memcpy(s.phi, leg[i].phi, sizeof s.phi); std::memcpy(s.dphi, leg[i].dphi, sizeof s.dphi); std::memcpy(s.c, leg[i].c, sizeof s.c);
    }
    if (!est->init_first_pose_flag) est->initFirstIMUPose(last_interval.data(), (int)last_interval.size());   // :452-453
    for (const vilo_sample &s : last_interval) est->processIMULeg(s);
    const int rc = est->processImage(feature.t, (int)feature.ids.size(), feature.ids.data(), feature.obs11.data(), feature.stereo.data());
    if (rc != VILO_OK) return rc < 0 ? rc : -rc;
    prevTime = curTime;
    featureBuf.pop_front();
    ++done;
  }
  return done;
}

}  // namespace vilo

extern "C" {
using vilo::MeasurementProcessor;
void *vilo_mp_create(void *sw) { return new MeasurementProcessor((vilo::SlidingWindow *)sw); }
void vilo_mp_destroy(void *h) { delete (MeasurementProcessor *)h; }
namespace {
struct Busy {   // adds the time of one entry point to the processor's account
  MeasurementProcessor *p;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit Busy(void *h) : p((MeasurementProcessor *)h) {}
  ~Busy() { p->busy_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace
void vilo_mp_input_imu(void *h, double t, const double *acc, const double *gyr) { Busy b(h); b.p->inputIMU(t, acc, gyr); }
void vilo_mp_input_leg(void *h, double t, const double *phi, const double *dphi, const double *c) { Busy b(h); b.p->inputLeg(t, phi, dphi, c); }
void vilo_mp_input_sample(void *h, double t, const double *row35) {
  Busy b(h);
  b.p->inputIMU(t, row35 + 1, row35 + 4);
  b.p->inputLeg(t, row35 + 7, row35 + 19, row35 + 31);
}
int vilo_mp_input_feature(void *h, double t, int n, const int *ids, const double *obs11, const uint8_t *stereo) {
  Busy b(h);
  vilo::FeatureFrame f;
  f.t = t;
  f.ids.assign(ids, ids + n); f.obs11.assign(obs11, obs11 + 11 * (size_t)n); f.stereo.assign(stereo, stereo + n);
  return b.p->inputFeature(f);
}
int vilo_mp_process(void *h) { Busy b(h); return b.p->processMeasurements(); }
double vilo_mp_busy_ms(void *h) { return 1e-6 * (double)((MeasurementProcessor *)h)->busy_ns; }
int vilo_mp_queue_size(void *h) { return (int)((MeasurementProcessor *)h)->buf.size(); }
int vilo_mp_last_interval(void *h, vilo_sample *out, int max_n) {
  const std::vector<vilo_sample> &v = ((MeasurementProcessor *)h)->last_interval;
  if ((int)v.size() > max_n) return -1;
  std::memcpy(out, v.data(), sizeof(vilo_sample) * v.size());
  return (int)v.size();
}
}
