// Dependency-free reader (and a matching writer, used by the round-trip tests and by tools/run_sequence.py --write-bag) of ROS bag
// files, format 2.0, for the four message types the reference's node consumes — SURVEY 8(f) rank 4. What it stands in for:
//
//   src/main.cpp:415-437   the node's subscriptions: IMU_TOPIC (sensor_msgs/Imu) and LEG_TOPIC (sensor_msgs/JointState) through an
//                          ApproximateTime synchroniser, IMAGE0_TOPIC / IMAGE1_TOPIC (sensor_msgs/Image), /feature_tracker/feature
//                          (sensor_msgs/PointCloud)
//   src/main.cpp:255-330   sensor_callback: acc / gyr from the Imu message, joint positions and velocities from JointState position[0..11] /
//                          velocity[0..11], the planner's contact flags from velocity[12..15], the foot-force readings from effort[12..15];
//                          which of the three contact sources reaches Estimator::inputLeg is CONTACT_SENSOR_TYPE (0: the Kalman filter's
//                          estimate — src/kalmanFilter is an absent submodule: the planner's flags stand in for it —, 1: the planner's
//                          flags, 2: the foot forces)
//   src/main.cpp:204-234   feature_callback: one point per (feature, camera) with channels id, camera_id, p_u, p_v, velocity_x, velocity_y
//   src/main.cpp:54-90     img0_callback / img1_callback / getImageFromMsg (the images feed the feature tracker, which is out of scope:
//                          the reader deserialises them, nothing consumes the pixels)
//
// Format (http://wiki.ros.org/Bags/Format/2.0, restated): "#ROSBAG V2.0\n", then records <header_len u32><header><data_len u32><data>,
// little endian, a header being fields <field_len u32><name>=<value>. Records: bag header (op 3; index_pos, conn_count, chunk_count; padded
// to 4096 bytes), chunk (op 5; compression, size; data = connection (op 7) and message data (op 2) records), index data (op 4) after each
// chunk, and at index_pos the connection records again + one chunk info (op 6) per chunk. The reader walks the chunks in file order and
// needs neither index; chunks compressed with bz2 / lz4 (rosbag record --bz2 / --lz4) are decompressed with the system's libbz2 / liblz4,
// loaded at run time — without them such a bag is refused (VILO_BAG_ERR_COMPRESSED). Messages
// are ROS 1 serialisation: little endian, strings and arrays prefixed with a u32 count.
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

namespace vilo {

enum BagMsgKind { BAG_IMU = 0, BAG_JOINT_STATE = 1, BAG_IMAGE = 2, BAG_POINT_CLOUD = 3, BAG_OTHER = 4 };

struct RosHeader {
  uint32_t seq = 0, secs = 0, nsecs = 0;
  std::string frame_id;
  double toSec() const { return (double)secs + 1e-9 * (double)nsecs; }   // ros::Time::toSec()
};
struct ImuMsg {   // sensor_msgs/Imu
  RosHeader header;
  double orientation[4] = {0, 0, 0, 1}, orientation_covariance[9] = {0};
  double angular_velocity[3] = {0}, angular_velocity_covariance[9] = {0};
  double linear_acceleration[3] = {0}, linear_acceleration_covariance[9] = {0};
};
struct JointStateMsg {   // sensor_msgs/JointState
  RosHeader header;
  std::vector<std::string> name;
  std::vector<double> position, velocity, effort;
};
struct ImageMsg {   // sensor_msgs/Image
  RosHeader header;
  uint32_t height = 0, width = 0, step = 0;
  std::string encoding;
  uint8_t is_bigendian = 0;
  std::vector<uint8_t> data;
};
struct PointCloudMsg {   // sensor_msgs/PointCloud: geometry_msgs/Point32[] points, ChannelFloat32[] channels
  RosHeader header;
  std::vector<float> points;   // x y z per point
  std::vector<std::string> channel_name;
  std::vector<std::vector<float>> channel_values;
};

void serialize(const ImuMsg &m, std::vector<uint8_t> *out);
void serialize(const JointStateMsg &m, std::vector<uint8_t> *out);
void serialize(const ImageMsg &m, std::vector<uint8_t> *out);
void serialize(const PointCloudMsg &m, std::vector<uint8_t> *out);
// false: the buffer is not a message of that type (ran past its end, or bytes are left over)
bool deserialize(const uint8_t *p, size_t n, ImuMsg *m);
bool deserialize(const uint8_t *p, size_t n, JointStateMsg *m);
bool deserialize(const uint8_t *p, size_t n, ImageMsg *m);
bool deserialize(const uint8_t *p, size_t n, PointCloudMsg *m);

const char *bag_type_name(int kind);     // "sensor_msgs/Imu", ...
const char *bag_type_md5(int kind);      // the md5sum rosbag records for the type's definition
const char *bag_type_definition(int kind);

struct BagMessage {
  std::string topic, type;
  uint32_t secs = 0, nsecs = 0;   // the record's time (when the message was received by the recorder)
  std::vector<uint8_t> data;      // serialised message
  int kind() const;
};

class BagWriter {
 public:
  ~BagWriter() { close(); }
  bool open(const char *path, size_t chunk_threshold = 768 * 1024);
  // "none" (default), "bz2" or "lz4": how the chunks written from now on are compressed (rosbag record --bz2 / --lz4). VILO_BAG_OK,
  // VILO_BAG_ERR_COMPRESSED if the system has no libbz2 / liblz4 to load, VILO_BAG_ERR_FORMAT for another name.
  int setCompression(const std::string &name);
  // messages may be written in any order of topics; a topic's type is fixed by its first message
  bool write(const std::string &topic, int kind, uint32_t secs, uint32_t nsecs, const std::vector<uint8_t> &data);
  bool close();   // flushes the last chunk, writes the index section and the bag header

 private:
  struct Conn { uint32_t id; std::string topic; int kind; };
  struct ChunkInfo { uint64_t pos, t0, t1; std::map<uint32_t, uint32_t> counts; };
  struct IndexEntry { uint64_t time; uint32_t offset; };
  void connection_record(const Conn &c, std::vector<uint8_t> *out) const;
  bool flush_chunk();
  FILE *f_ = nullptr;
  size_t threshold_ = 0;
  std::string compression_ = "none";
  std::vector<Conn> conns_;
  std::map<std::string, uint32_t> by_topic_;
  std::vector<uint8_t> chunk_;
  std::map<uint32_t, std::vector<IndexEntry>> chunk_index_;
  std::map<uint32_t, bool> conn_in_chunk_;   // connection record written (in the chunk of its first message)
  uint64_t chunk_t0_ = 0, chunk_t1_ = 0;
  std::vector<ChunkInfo> infos_;
};

enum { VILO_BAG_OK = 0, VILO_BAG_END = 1, VILO_BAG_ERR_IO = -1, VILO_BAG_ERR_FORMAT = -2, VILO_BAG_ERR_COMPRESSED = -3 };

class BagReader {
 public:
  ~BagReader() { close(); }
  int open(const char *path);   // VILO_BAG_OK or an error
  int next(BagMessage *m);      // VILO_BAG_OK, VILO_BAG_END or an error; messages come in file order
  void close();
  uint32_t conn_count = 0, chunk_count = 0;   // from the bag header
  uint64_t index_pos = 0;

 private:
  int load_next_chunk();
  FILE *f_ = nullptr;
  long file_size_ = 0;          // taken once at open (fstat): bounds every length a record states
  std::vector<uint8_t> chunk_;
  size_t pos_ = 0;
  std::map<uint32_t, std::pair<std::string, std::string>> conns_;   // conn -> (topic, type)
};

}  // namespace vilo

// ---- C entry points (cerberus_amd/rosbag.py) ----
extern "C" {
typedef struct vilo_bag_msg {
  int32_t kind;                    // BagMsgKind
  uint32_t rec_secs, rec_nsecs;    // record time
  uint32_t seq, secs, nsecs;       // header (kinds 0 .. 3)
  char topic[256], type[64], frame_id[64];
  // Imu
  double orientation[4], angular_velocity[3], linear_acceleration[3];
  // JointState: the first 32 entries of each array (the reference's messages carry 16)
  int32_t n_position, n_velocity, n_effort;
  double position[32], velocity[32], effort[32];
  // Image
  uint32_t height, width, step;
  char encoding[32];
  int32_t is_bigendian;
  // PointCloud
  int32_t n_points, n_channels;
  // variable-size parts, valid until the next call on the same reader: Image data; PointCloud points (xyz float32) and channel c's values at
  // channels + c * n_points
  const uint8_t *data;
  uint32_t data_len;
  const float *points;
  const float *channels;
  char channel_names[16][32];
} vilo_bag_msg;

void *vilo_bag_writer_open(const char *path, int chunk_threshold_bytes);
int vilo_bag_write_imu(void *h, const char *topic, uint32_t seq, uint32_t secs, uint32_t nsecs, const char *frame_id, const double *acc3, const double *gyr3);
// position / velocity / effort: n doubles each (the reference: 12 joints + 4 feet; velocity[12 ..] the planner's contact flags, effort[12 ..] the foot forces)
int vilo_bag_write_joint_state(void *h, const char *topic, uint32_t seq, uint32_t secs, uint32_t nsecs, int n, const double *position, const double *velocity,
                               const double *effort);
int vilo_bag_write_image(void *h, const char *topic, uint32_t seq, uint32_t secs, uint32_t nsecs, const char *frame_id, uint32_t height, uint32_t width,
                         const char *encoding, uint32_t step, const uint8_t *data);
// channels: n_channels x n_points float32, channel-major
int vilo_bag_write_point_cloud(void *h, const char *topic, uint32_t seq, uint32_t secs, uint32_t nsecs, int n_points, const float *xyz, int n_channels,
                               const char *const *channel_names, const float *channels);
int vilo_bag_writer_set_compression(void *h, const char *name);   // BagWriter::setCompression
int vilo_bag_writer_close(void *h);   // 0 ok; frees the handle

void *vilo_bag_reader_open(const char *path, int *rc);
int vilo_bag_reader_info(void *h, uint32_t *conn_count, uint32_t *chunk_count, uint64_t *index_pos);
int vilo_bag_reader_next(void *h, vilo_bag_msg *out);   // 0 a message, 1 end of the bag, < 0 an error
void vilo_bag_reader_close(void *h);
}
