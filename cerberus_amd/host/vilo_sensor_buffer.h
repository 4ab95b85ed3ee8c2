// Sensor-interval extraction: the queues between the sensor callbacks and Estimator::processImage, restated
// without ROS types or threads.
//
//   reference (src/estimator/estimator.cpp)                       here
//   inputIMU :255-273, inputLeg :275-289                          SensorBuffer::inputIMU, inputLeg        (fastPredictIMU / publishing: not restated)
//   inputFeature :291-299                                         MeasurementProcessor::inputFeature
//   IMUAvailable :340-346                                         SensorBuffer::IMUAvailable
//   getIMUAndLegInterval :349-397                                 SensorBuffer::getIMUAndLegInterval
//   processMeasurements :400-521 (USE_LEG && USE_IMU branch)      MeasurementProcessor::processMeasurements: dt rule :456-462, first-pose
//                                                                 initialisation :452-453, processIMULeg per sample, processImage
//
// Parity status: restated from the source; the reference's Estimator cannot be compiled here, so "parity unpinned" except through
// tests/test_sensor_buffer.py's statement of the queue semantics (which samples an interval holds, which stay queued, each dt).
#pragma once
#include <deque>
#include <utility>
#include <vector>

#include "vilo_sliding_window.h"

namespace vilo {

struct ImuMsg { double t, acc[3], gyr[3]; };
struct LegMsg { double t, phi[12], dphi[12], c[4]; };

class SensorBuffer {
 public:
  void clear() { imu_.clear(); leg_.clear(); }
  void inputIMU(double t, const double acc[3], const double gyr[3]);
  void inputLeg(double t, const double phi[12], const double dphi[12], const double c[4]);
  bool IMUAvailable(double t) const { return !imu_.empty() && t <= imu_.back().t; }
  // Messages with t0 < stamp < t1 and the first one at or after t1 (that one stays queued: it also opens the next interval).
  // Leg messages are popped in lockstep with the IMU ones ("assume leg measurement aligns with IMU measurement very well", :348).
  // false: nothing queued, or the queue does not reach t1 yet ("wait for imu and leg").
  bool getIMUAndLegInterval(double t0, double t1, std::vector<ImuMsg> *imu, std::vector<LegMsg> *leg);
  size_t size() const { return imu_.size(); }

 private:
  std::deque<ImuMsg> imu_;
  std::deque<LegMsg> leg_;
};

// One image worth of tracked features in vilo_window_desc::obs order
struct FeatureFrame {
  double t;
  std::vector<int> ids;
  std::vector<double> obs11;
  std::vector<uint8_t> stereo;
};

class MeasurementProcessor {
 public:
  explicit MeasurementProcessor(SlidingWindow *estimator) : est(estimator) {}
  void inputIMU(double t, const double acc[3], const double gyr[3]) { buf.inputIMU(t, acc, gyr); }
  void inputLeg(double t, const double phi[12], const double dphi[12], const double c[4]) { buf.inputLeg(t, phi, dphi, c); }
  // Queues the frame and processes every queued frame whose interval is complete (MULTIPLE_THREAD = 0 behaviour, :297-298).
  // Returns the number of images processed, or a negative vilo_status.
  int inputFeature(const FeatureFrame &f);
  int processMeasurements();

  SensorBuffer buf;
  std::deque<FeatureFrame> featureBuf;
  SlidingWindow *est;
  double prevTime = -1.0, curTime = 0.0;   // clearState :43-44
  std::vector<vilo_sample> last_interval;  // the samples handed to processIMULeg for the newest image (dt filled in)
  long long busy_ns = 0;                   // wall time spent inside the C entry points below (vilo_mp_busy_ms): what a C++ node pays per message
};

}  // namespace vilo

extern "C" {
void *vilo_mp_create(void *sliding_window);
void vilo_mp_destroy(void *h);
void vilo_mp_input_imu(void *h, double t, const double *acc, const double *gyr);
void vilo_mp_input_leg(void *h, double t, const double *phi, const double *dphi, const double *c);
// returns the number of images processed (0: waiting for IMU/leg data up to the image time), < 0: vilo_status
int vilo_mp_input_feature(void *h, double t, int n, const int *ids, const double *obs11, const uint8_t *stereo);
// processMeasurements() again (a queued image whose interval has been completed by later messages)
int vilo_mp_process(void *h);
// introspection for tests: queue length, and the samples (with dt) of the newest processed interval
int vilo_mp_queue_size(void *h);
// wall time spent so far inside vilo_mp_input_imu / _input_leg / _input_sample / _input_feature / _process of this processor (ms)
double vilo_mp_busy_ms(void *h);
// one sample row as the synthetic stream lays it out (vilo_sample: dt acc gyr phi dphi c): inputIMU + inputLeg with one call
void vilo_mp_input_sample(void *h, double t, const double *row35);
int vilo_mp_last_interval(void *h, vilo_sample *out, int max_n);
}
