// ROS bag (format 2.0) reader / writer and ROS 1 (de)serialisation of the messages the reference's node consumes: see vilo_rosbag.h.
#include "vilo_rosbag.h"

#include <dlfcn.h>
#include <sys/stat.h>

#include <new>

#include <cstring>

namespace vilo {

// ---------------------------------------------------------------------------------------------------------------------------------
// little-endian primitives
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
void put_u32(std::vector<uint8_t> *o, uint32_t v) { for (int i = 0; i < 4; ++i) o->push_back((uint8_t)(v >> (8 * i))); }
void put_u64(std::vector<uint8_t> *o, uint64_t v) { for (int i = 0; i < 8; ++i) o->push_back((uint8_t)(v >> (8 * i))); }
void put_bytes(std::vector<uint8_t> *o, const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; o->insert(o->end(), b, b + n); }
void put_f64(std::vector<uint8_t> *o, double v) { uint64_t u; std::memcpy(&u, &v, 8); put_u64(o, u); }
void put_f32(std::vector<uint8_t> *o, float v) { uint32_t u; std::memcpy(&u, &v, 4); put_u32(o, u); }
void put_str(std::vector<uint8_t> *o, const std::string &s) { put_u32(o, (uint32_t)s.size()); put_bytes(o, s.data(), s.size()); }
uint32_t get_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t get_u64(const uint8_t *p) { return (uint64_t)get_u32(p) | ((uint64_t)get_u32(p + 4) << 32); }

// a cursor over a serialised message that remembers running past the end
struct Cur {
  const uint8_t *p; size_t n, at = 0; bool ok = true;
  Cur(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
  bool need(size_t k) { if (!ok || n - at < k) { ok = false; return false; } return true; }
  uint8_t u8() { if (!need(1)) return 0; return p[at++]; }
  uint32_t u32() { if (!need(4)) return 0; const uint32_t v = get_u32(p + at); at += 4; return v; }
  double f64() { if (!need(8)) return 0.0; const uint64_t u = get_u64(p + at); at += 8; double v; std::memcpy(&v, &u, 8); return v; }
  float f32() { if (!need(4)) return 0.0f; const uint32_t u = get_u32(p + at); at += 4; float v; std::memcpy(&v, &u, 4); return v; }
  std::string str() { const uint32_t k = u32(); if (!need(k)) return std::string(); std::string s((const char *)p + at, k); at += k; return s; }
  void f64s(double *d, int k) { for (int i = 0; i < k; ++i) d[i] = f64(); }
  bool done() const { return ok && at == n; }
};
void put_header(std::vector<uint8_t> *o, const RosHeader &h) { put_u32(o, h.seq); put_u32(o, h.secs); put_u32(o, h.nsecs); put_str(o, h.frame_id); }
void get_header(Cur &c, RosHeader *h) { h->seq = c.u32(); h->secs = c.u32(); h->nsecs = c.u32(); h->frame_id = c.str(); }
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// messages
// ---------------------------------------------------------------------------------------------------------------------------------
void serialize(const ImuMsg &m, std::vector<uint8_t> *o) {
  put_header(o, m.header);
  for (int i = 0; i < 4; ++i) put_f64(o, m.orientation[i]);
  for (int i = 0; i < 9; ++i) put_f64(o, m.orientation_covariance[i]);
  for (int i = 0; i < 3; ++i) put_f64(o, m.angular_velocity[i]);
  for (int i = 0; i < 9; ++i) put_f64(o, m.angular_velocity_covariance[i]);
  for (int i = 0; i < 3; ++i) put_f64(o, m.linear_acceleration[i]);
  for (int i = 0; i < 9; ++i) put_f64(o, m.linear_acceleration_covariance[i]);
}
bool deserialize(const uint8_t *p, size_t n, ImuMsg *m) {
  Cur c(p, n);
  get_header(c, &m->header);
  c.f64s(m->orientation, 4); c.f64s(m->orientation_covariance, 9);
  c.f64s(m->angular_velocity, 3); c.f64s(m->angular_velocity_covariance, 9);
  c.f64s(m->linear_acceleration, 3); c.f64s(m->linear_acceleration_covariance, 9);
  return c.done();
}
void serialize(const JointStateMsg &m, std::vector<uint8_t> *o) {
  put_header(o, m.header);
  put_u32(o, (uint32_t)m.name.size());
  for (const std::string &s : m.name) put_str(o, s);
  for (const std::vector<double> *v : {&m.position, &m.velocity, &m.effort}) {
    put_u32(o, (uint32_t)v->size());
    for (double d : *v) put_f64(o, d);
  }
}
bool deserialize(const uint8_t *p, size_t n, JointStateMsg *m) {
  Cur c(p, n);
  get_header(c, &m->header);
  const uint32_t nn = c.u32();
  m->name.clear();
  for (uint32_t i = 0; i < nn && c.ok; ++i) m->name.push_back(c.str());
  for (std::vector<double> *v : {&m->position, &m->velocity, &m->effort}) {
    const uint32_t k = c.u32();
    v->clear();
    if (!c.need((size_t)k * 8)) return false;
    for (uint32_t i = 0; i < k; ++i) v->push_back(c.f64());
  }
  return c.done();
}
void serialize(const ImageMsg &m, std::vector<uint8_t> *o) {
  put_header(o, m.header);
  put_u32(o, m.height); put_u32(o, m.width); put_str(o, m.encoding); o->push_back(m.is_bigendian); put_u32(o, m.step);
  put_u32(o, (uint32_t)m.data.size()); put_bytes(o, m.data.data(), m.data.size());
}
bool deserialize(const uint8_t *p, size_t n, ImageMsg *m) {
  Cur c(p, n);
  get_header(c, &m->header);
  m->height = c.u32(); m->width = c.u32(); m->encoding = c.str(); m->is_bigendian = c.u8(); m->step = c.u32();
  const uint32_t k = c.u32();
  if (!c.need(k)) return false;
  m->data.assign(c.p + c.at, c.p + c.at + k); c.at += k;
  return c.done();
}
void serialize(const PointCloudMsg &m, std::vector<uint8_t> *o) {
  put_header(o, m.header);
  put_u32(o, (uint32_t)(m.points.size() / 3));
  for (float v : m.points) put_f32(o, v);
  put_u32(o, (uint32_t)m.channel_name.size());
  for (size_t ch = 0; ch < m.channel_name.size(); ++ch) {
    put_str(o, m.channel_name[ch]);
    put_u32(o, (uint32_t)m.channel_values[ch].size());
    for (float v : m.channel_values[ch]) put_f32(o, v);
  }
}
bool deserialize(const uint8_t *p, size_t n, PointCloudMsg *m) {
  Cur c(p, n);
  get_header(c, &m->header);
  const uint32_t np = c.u32();
  if (!c.need((size_t)np * 12)) return false;
  m->points.resize((size_t)np * 3);
  for (float &v : m->points) v = c.f32();
  const uint32_t nc = c.u32();
  m->channel_name.clear(); m->channel_values.clear();
  for (uint32_t ch = 0; ch < nc && c.ok; ++ch) {
    m->channel_name.push_back(c.str());
    const uint32_t k = c.u32();
    if (!c.need((size_t)k * 4)) return false;
    std::vector<float> v(k);
    for (float &x : v) x = c.f32();
    m->channel_values.push_back(std::move(v));
  }
  return c.done();
}

const char *bag_type_name(int kind) {
  switch (kind) {
    case BAG_IMU: return "sensor_msgs/Imu";
    case BAG_JOINT_STATE: return "sensor_msgs/JointState";
    case BAG_IMAGE: return "sensor_msgs/Image";
    case BAG_POINT_CLOUD: return "sensor_msgs/PointCloud";
    default: return "";
  }
}
const char *bag_type_md5(int kind) {   // as published with the message packages (common_msgs); the reader does not check them
  switch (kind) {
    case BAG_IMU: return "6a62c6daae103f4ff57a132d6f95cec2";
    case BAG_JOINT_STATE: return "3066dcd76a6cfaef579bd0f34173e9fd";
    case BAG_IMAGE: return "060021388200f6f0f447d0fcd9c64743";
    case BAG_POINT_CLOUD: return "d8e9c3f5afbdd8a130fd1d2763945fca";
    default: return "";
  }
}
// Message definitions as rosbag stores them: the top-level text followed by one "MSG: <package>/<type>" section per dependency, separated by
// lines of 80 '='. rosbag's Python API, rqt_bag and `rostopic echo -b` build the message classes from this text (the md5sums above are what
// a C++ subscriber checks); the comments of the .msg files are left out, which neither changes the classes nor the md5sums.
#define VILO_MSG_SEP "================================================================================\n"
#define VILO_MSG_HEADER VILO_MSG_SEP "MSG: std_msgs/Header\nuint32 seq\ntime stamp\nstring frame_id\n"
const char *bag_type_definition(int kind) {
  switch (kind) {
    case BAG_IMU:
      return "Header header\ngeometry_msgs/Quaternion orientation\nfloat64[9] orientation_covariance\ngeometry_msgs/Vector3 angular_velocity\n"
             "float64[9] angular_velocity_covariance\ngeometry_msgs/Vector3 linear_acceleration\nfloat64[9] linear_acceleration_covariance\n"
             VILO_MSG_HEADER
             VILO_MSG_SEP "MSG: geometry_msgs/Quaternion\nfloat64 x\nfloat64 y\nfloat64 z\nfloat64 w\n"
             VILO_MSG_SEP "MSG: geometry_msgs/Vector3\nfloat64 x\nfloat64 y\nfloat64 z\n";
    case BAG_JOINT_STATE: return "Header header\nstring[] name\nfloat64[] position\nfloat64[] velocity\nfloat64[] effort\n" VILO_MSG_HEADER;
    case BAG_IMAGE: return "Header header\nuint32 height\nuint32 width\nstring encoding\nuint8 is_bigendian\nuint32 step\nuint8[] data\n" VILO_MSG_HEADER;
    case BAG_POINT_CLOUD:
      return "Header header\ngeometry_msgs/Point32[] points\nChannelFloat32[] channels\n"
             VILO_MSG_HEADER
             VILO_MSG_SEP "MSG: geometry_msgs/Point32\nfloat32 x\nfloat32 y\nfloat32 z\n"
             VILO_MSG_SEP "MSG: sensor_msgs/ChannelFloat32\nstring name\nfloat32[] values\n";
    default: return "";
  }
}
int BagMessage::kind() const {
  for (int k = 0; k < 4; ++k) if (type == bag_type_name(k)) return k;
  return BAG_OTHER;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// records
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
void put_field(std::vector<uint8_t> *h, const char *name, const void *val, size_t n) {
  const size_t ln = std::strlen(name);
  put_u32(h, (uint32_t)(ln + 1 + n));
  put_bytes(h, name, ln);
  h->push_back('=');
  put_bytes(h, val, n);
}
void put_field_u8(std::vector<uint8_t> *h, const char *name, uint8_t v) { put_field(h, name, &v, 1); }
void put_field_u32(std::vector<uint8_t> *h, const char *name, uint32_t v) { std::vector<uint8_t> b; put_u32(&b, v); put_field(h, name, b.data(), 4); }
void put_field_u64(std::vector<uint8_t> *h, const char *name, uint64_t v) { std::vector<uint8_t> b; put_u64(&b, v); put_field(h, name, b.data(), 8); }
void put_field_str(std::vector<uint8_t> *h, const char *name, const std::string &s) { put_field(h, name, s.data(), s.size()); }
void put_record(std::vector<uint8_t> *o, const std::vector<uint8_t> &header, const std::vector<uint8_t> &data) {
  put_u32(o, (uint32_t)header.size()); put_bytes(o, header.data(), header.size());
  put_u32(o, (uint32_t)data.size()); put_bytes(o, data.data(), data.size());
}
// fields of a record header: name -> value bytes. false: malformed
bool parse_fields(const uint8_t *p, size_t n, std::map<std::string, std::string> *out) {
  size_t at = 0;
  out->clear();
  while (at < n) {
    if (n - at < 4) return false;
    const uint32_t len = get_u32(p + at); at += 4;
    if (len == 0 || n - at < len) return false;
    const uint8_t *eq = (const uint8_t *)std::memchr(p + at, '=', len);
    if (!eq) return false;
    (*out)[std::string((const char *)p + at, eq - (p + at))] = std::string((const char *)eq + 1, len - (eq + 1 - (p + at)));
    at += len;
  }
  return true;
}
bool field_u32(const std::map<std::string, std::string> &f, const char *name, uint32_t *v) {
  auto it = f.find(name);
  if (it == f.end() || it->second.size() != 4) return false;
  *v = get_u32((const uint8_t *)it->second.data());
  return true;
}
bool field_u64(const std::map<std::string, std::string> &f, const char *name, uint64_t *v) {
  auto it = f.find(name);
  if (it == f.end() || it->second.size() != 8) return false;
  *v = get_u64((const uint8_t *)it->second.data());
  return true;
}
int field_op(const std::map<std::string, std::string> &f) {
  auto it = f.find("op");
  return (it == f.end() || it->second.size() != 1) ? -1 : (int)(uint8_t)it->second[0];
}
const char MAGIC[] = "#ROSBAG V2.0\n";
const size_t MAGIC_N = 13, BAG_HEADER_RECORD = 4096;
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// writer
// ---------------------------------------------------------------------------------------------------------------------------------
bool BagWriter::open(const char *path, size_t chunk_threshold) {
  close();
  f_ = std::fopen(path, "wb");
  if (!f_) return false;
  threshold_ = chunk_threshold;
  compression_ = "none";
  conns_.clear(); by_topic_.clear(); chunk_.clear(); chunk_index_.clear(); conn_in_chunk_.clear(); infos_.clear();
  std::vector<uint8_t> blank(MAGIC_N + BAG_HEADER_RECORD, (uint8_t)' ');   // the header record is written by close()
  std::memcpy(blank.data(), MAGIC, MAGIC_N);
  return std::fwrite(blank.data(), 1, blank.size(), f_) == blank.size();
}
// ---- chunk compression: rosbag writes chunks "none", "bz2" (one bz2 stream of the chunk's records) or "lz4" (roslz4: one LZ4 frame,
// format 1.4.1 — magic 0x184D2204, independent blocks, content checksum). Neither library has headers in this image and the product must
// not depend on them, so libbz2.so.1.0 / liblz4.so.1 are looked up at run time (dlopen) and used through their stable C ABI
// (BZ2_bzBuffToBuff{Decompress,Compress}; LZ4F_* of the frame API); without them such chunks are refused as before (VILO_BAG_ERR_COMPRESSED). ----
namespace {
struct Bz2Api {
  int (*decompress)(char *, unsigned *, char *, unsigned, int, int) = nullptr;
  int (*compress)(char *, unsigned *, char *, unsigned, int, int, int) = nullptr;
};
const Bz2Api *bz2_api() {
  static const Bz2Api api = [] {
    Bz2Api a;
    void *h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
    if (h) {
      a.decompress = (decltype(a.decompress))dlsym(h, "BZ2_bzBuffToBuffDecompress");
      a.compress = (decltype(a.compress))dlsym(h, "BZ2_bzBuffToBuffCompress");
    }
    return a;
  }();
  return (api.decompress && api.compress) ? &api : nullptr;
}
// lz4frame.h of liblz4 1.7 .. 1.9 (the structs are part of the library's ABI)
struct Lz4FrameInfo { int blockSizeID, blockMode, contentChecksumFlag, frameType; unsigned long long contentSize; unsigned dictID; int blockChecksumFlag; };
struct Lz4Preferences { Lz4FrameInfo frameInfo; int compressionLevel; unsigned autoFlush, favorDecSpeed, reserved[3]; };
struct Lz4Api {
  size_t (*create_dctx)(void **, unsigned) = nullptr;
  size_t (*free_dctx)(void *) = nullptr;
  size_t (*decompress)(void *, void *, size_t *, const void *, size_t *, const void *) = nullptr;
  unsigned (*is_error)(size_t) = nullptr;
  size_t (*bound)(size_t, const Lz4Preferences *) = nullptr;
  size_t (*compress_frame)(void *, size_t, const void *, size_t, const Lz4Preferences *) = nullptr;
};
const Lz4Api *lz4_api() {
  static const Lz4Api api = [] {
    Lz4Api a;
    void *h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL);
    if (h) {
      a.create_dctx = (decltype(a.create_dctx))dlsym(h, "LZ4F_createDecompressionContext");
      a.free_dctx = (decltype(a.free_dctx))dlsym(h, "LZ4F_freeDecompressionContext");
      a.decompress = (decltype(a.decompress))dlsym(h, "LZ4F_decompress");
      a.is_error = (decltype(a.is_error))dlsym(h, "LZ4F_isError");
      a.bound = (decltype(a.bound))dlsym(h, "LZ4F_compressFrameBound");
      a.compress_frame = (decltype(a.compress_frame))dlsym(h, "LZ4F_compressFrame");
    }
    return a;
  }();
  return (api.create_dctx && api.free_dctx && api.decompress && api.is_error && api.bound && api.compress_frame) ? &api : nullptr;
}
// VILO_BAG_OK, VILO_BAG_ERR_COMPRESSED (no library for this compression) or VILO_BAG_ERR_FORMAT (the stream is not what the header says)
int chunk_decompress(const std::string &compression, const std::vector<uint8_t> &src, uint32_t size, std::vector<uint8_t> *out) {
  // `size` is the header's word for the uncompressed length (untrusted), checked before anything is allocated. rosbag's chunks are 768 KiB
  // by default (`--chunksize` is in KiB and rarely beyond tens of MiB): 256 MiB is the absolute cap for either codec. An LZ4 block cannot
  // expand more than 255-fold (one length byte per 255 repeated bytes), so its claim is also held to the compressed bytes it comes with;
  // bz2 has no such bound worth using — 768 KiB of zeros are 46 bytes, three black 640 x 480 images 160 — and gets the cap only.
  if (size > (1u << 28)) return VILO_BAG_ERR_FORMAT;
  if (compression == "lz4" && (uint64_t)size > 256ull * (uint64_t)src.size() + (1u << 16)) return VILO_BAG_ERR_FORMAT;
  out->assign(size, 0);
  if (compression == "bz2") {
    const Bz2Api *a = bz2_api();
    if (!a) return VILO_BAG_ERR_COMPRESSED;
    unsigned n = size;
    const int rc = a->decompress((char *)out->data(), &n, (char *)src.data(), (unsigned)src.size(), 0, 0);
    return (rc == 0 && n == size) ? VILO_BAG_OK : VILO_BAG_ERR_FORMAT;
  }
  if (compression == "lz4") {
    const Lz4Api *a = lz4_api();
    if (!a) return VILO_BAG_ERR_COMPRESSED;
    void *ctx = nullptr;
    if (a->is_error(a->create_dctx(&ctx, 100 /* LZ4F_VERSION */)) || !ctx) return VILO_BAG_ERR_COMPRESSED;
    size_t in_pos = 0, out_pos = 0, hint = 1;
    bool ok = true;
    while (ok && in_pos < src.size()) {
      size_t dn = size - out_pos, sn = src.size() - in_pos;
      hint = a->decompress(ctx, out->data() + out_pos, &dn, src.data() + in_pos, &sn, nullptr);
      if (a->is_error(hint) || (dn == 0 && sn == 0)) { ok = false; break; }
      in_pos += sn; out_pos += dn;
      if (hint == 0) break;   // the frame is complete
    }
    a->free_dctx(ctx);
    return (ok && hint == 0 && out_pos == size && in_pos == src.size()) ? VILO_BAG_OK : VILO_BAG_ERR_FORMAT;
  }
  return VILO_BAG_ERR_COMPRESSED;
}
bool chunk_compress(const std::string &compression, const std::vector<uint8_t> &src, std::vector<uint8_t> *out) {
  if (compression == "bz2") {
    const Bz2Api *a = bz2_api();
    if (!a) return false;
    unsigned n = (unsigned)(src.size() + src.size() / 100 + 600);
    out->assign(n, 0);
    if (a->compress((char *)out->data(), &n, (char *)src.data(), (unsigned)src.size(), 9, 0, 30) != 0) return false;   // (rosbag: block size 9)
    out->resize(n);
    return true;
  }
  if (compression == "lz4") {
    const Lz4Api *a = lz4_api();
    if (!a) return false;
    Lz4Preferences pr;
    std::memset(&pr, 0, sizeof(pr));
    pr.frameInfo.blockSizeID = 7;          // LZ4F_max4MB
    pr.frameInfo.blockMode = 1;            // LZ4F_blockIndependent (roslz4 writes and expects independent blocks)
    pr.frameInfo.contentChecksumFlag = 1;  // the stream checksum roslz4 appends
    out->assign(a->bound(src.size(), &pr), 0);
    const size_t n = a->compress_frame(out->data(), out->size(), src.data(), src.size(), &pr);
    if (a->is_error(n)) return false;
    out->resize(n);
    return true;
  }
  return false;
}
}  // namespace

int BagWriter::setCompression(const std::string &name) {
  if (name != "none" && name != "bz2" && name != "lz4") return VILO_BAG_ERR_FORMAT;
  if ((name == "bz2" && !bz2_api()) || (name == "lz4" && !lz4_api())) return VILO_BAG_ERR_COMPRESSED;
  if (!chunk_.empty() && name != compression_ && !flush_chunk()) return VILO_BAG_ERR_IO;   // (a chunk has one compression)
  compression_ = name;
  return VILO_BAG_OK;
}
void BagWriter::connection_record(const Conn &c, std::vector<uint8_t> *out) const {
  std::vector<uint8_t> h, d;
  put_field_u8(&h, "op", 0x07); put_field_u32(&h, "conn", c.id); put_field_str(&h, "topic", c.topic);
  put_field_str(&d, "topic", c.topic); put_field_str(&d, "type", bag_type_name(c.kind)); put_field_str(&d, "md5sum", bag_type_md5(c.kind));
  put_field_str(&d, "message_definition", bag_type_definition(c.kind));
  put_record(out, h, d);
}
bool BagWriter::write(const std::string &topic, int kind, uint32_t secs, uint32_t nsecs, const std::vector<uint8_t> &data) {
  if (!f_ || kind < 0 || kind > 3) return false;
  auto it = by_topic_.find(topic);
  if (it == by_topic_.end()) {
    conns_.push_back(Conn{(uint32_t)conns_.size(), topic, kind});
    it = by_topic_.emplace(topic, conns_.back().id).first;
  }
  const Conn &c = conns_[it->second];
  if (c.kind != kind) return false;
  if (!conn_in_chunk_[c.id]) { connection_record(c, &chunk_); conn_in_chunk_[c.id] = true; }   // (once per bag, in the chunk of the connection's first message: as rosbag does)
  const uint64_t t = (uint64_t)secs | ((uint64_t)nsecs << 32);
  const uint64_t tcmp = ((uint64_t)secs << 32) | nsecs;   // (ordering only)
  if (chunk_index_.empty()) { chunk_t0_ = tcmp; chunk_t1_ = tcmp; }   // (the chunk's first message)
  else { if (tcmp < chunk_t0_) chunk_t0_ = tcmp; if (tcmp > chunk_t1_) chunk_t1_ = tcmp; }
  chunk_index_[c.id].push_back(IndexEntry{t, (uint32_t)chunk_.size()});
  std::vector<uint8_t> h;
  put_field_u8(&h, "op", 0x02); put_field_u32(&h, "conn", c.id); put_field_u64(&h, "time", t);
  put_record(&chunk_, h, data);
  if (chunk_.size() >= threshold_) return flush_chunk();
  return true;
}
bool BagWriter::flush_chunk() {
  if (chunk_.empty()) return true;
  ChunkInfo info;
  info.pos = (uint64_t)std::ftell(f_);
  auto unpack = [](uint64_t tcmp) { return (tcmp >> 32) | ((tcmp & 0xffffffffULL) << 32); };   // back to secs (low) | nsecs (high)
  info.t0 = unpack(chunk_t0_); info.t1 = unpack(chunk_t1_);
  std::vector<uint8_t> out, h;
  put_field_u8(&h, "op", 0x05); put_field_str(&h, "compression", compression_); put_field_u32(&h, "size", (uint32_t)chunk_.size());   // (size: uncompressed)
  if (compression_ == "none") put_record(&out, h, chunk_);
  else {
    std::vector<uint8_t> packed;
    if (!chunk_compress(compression_, chunk_, &packed)) return false;
    put_record(&out, h, packed);
  }
  for (const auto &kv : chunk_index_) {
    if (kv.second.empty()) continue;
    std::vector<uint8_t> ih, id;
    put_field_u8(&ih, "op", 0x04); put_field_u32(&ih, "ver", 1); put_field_u32(&ih, "conn", kv.first); put_field_u32(&ih, "count", (uint32_t)kv.second.size());
    for (const IndexEntry &e : kv.second) { put_u64(&id, e.time); put_u32(&id, e.offset); }
    put_record(&out, ih, id);
    info.counts[kv.first] = (uint32_t)kv.second.size();
  }
  infos_.push_back(info);
  chunk_.clear(); chunk_index_.clear();
  return std::fwrite(out.data(), 1, out.size(), f_) == out.size();
}
bool BagWriter::close() {
  if (!f_) return true;
  bool ok = flush_chunk();
  const uint64_t index_pos = (uint64_t)std::ftell(f_);
  std::vector<uint8_t> out;
  for (const Conn &c : conns_) connection_record(c, &out);
  for (const ChunkInfo &ci : infos_) {
    std::vector<uint8_t> h, d;
    put_field_u8(&h, "op", 0x06); put_field_u32(&h, "ver", 1); put_field_u64(&h, "chunk_pos", ci.pos); put_field_u64(&h, "start_time", ci.t0);
    put_field_u64(&h, "end_time", ci.t1); put_field_u32(&h, "count", (uint32_t)ci.counts.size());
    for (const auto &kv : ci.counts) { put_u32(&d, kv.first); put_u32(&d, kv.second); }
    put_record(&out, h, d);
  }
  ok = ok && std::fwrite(out.data(), 1, out.size(), f_) == out.size();
  // the bag header record, padded with spaces to 4096 bytes
  std::vector<uint8_t> h, rec;
  put_field_u8(&h, "op", 0x03); put_field_u64(&h, "index_pos", index_pos); put_field_u32(&h, "conn_count", (uint32_t)conns_.size());
  put_field_u32(&h, "chunk_count", (uint32_t)infos_.size());
  const std::vector<uint8_t> pad(BAG_HEADER_RECORD - 8 - h.size(), (uint8_t)' ');
  put_record(&rec, h, pad);
  ok = ok && std::fseek(f_, (long)MAGIC_N, SEEK_SET) == 0 && std::fwrite(rec.data(), 1, rec.size(), f_) == rec.size();
  ok = (std::fclose(f_) == 0) && ok;
  f_ = nullptr;
  return ok;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// reader
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
// one top-level record from the file. 1: got one, 0: clean end of file, < 0: error
int read_record(FILE *f, long file_size, std::vector<uint8_t> *header, std::vector<uint8_t> *data) {
  uint8_t b[4];
  const size_t got = std::fread(b, 1, 4, f);
  if (got == 0) return 0;
  if (got != 4) return VILO_BAG_ERR_FORMAT;
  const uint32_t hl = get_u32(b);
  if (hl > (1u << 26)) return VILO_BAG_ERR_FORMAT;
  header->resize(hl);
  if (hl && std::fread(header->data(), 1, hl, f) != hl) return VILO_BAG_ERR_FORMAT;
  if (std::fread(b, 1, 4, f) != 4) return VILO_BAG_ERR_FORMAT;
  const uint32_t dl = get_u32(b);
  // the length comes from the (untrusted) file: never allocate more than the file still holds (its size was taken once, at open: a seek
  // to the end and back per record would throw away the read buffer every time)
  const long here = std::ftell(f);
  if (here < 0) return VILO_BAG_ERR_IO;
  if (here > file_size || (unsigned long)dl > (unsigned long)(file_size - here)) return VILO_BAG_ERR_FORMAT;
  data->resize(dl);
  if (dl && std::fread(data->data(), 1, dl, f) != dl) return VILO_BAG_ERR_FORMAT;
  return 1;
}
bool register_connection(const std::map<std::string, std::string> &hf, const std::vector<uint8_t> &data,
                         std::map<uint32_t, std::pair<std::string, std::string>> *conns) {
  uint32_t id;
  if (!field_u32(hf, "conn", &id)) return false;
  std::map<std::string, std::string> df;
  if (!parse_fields(data.data(), data.size(), &df)) return false;
  auto t = hf.find("topic");
  (*conns)[id] = std::make_pair(t != hf.end() ? t->second : df["topic"], df["type"]);
  return true;
}
}  // namespace

int BagReader::open(const char *path) {
  close();
  f_ = std::fopen(path, "rb");
  if (!f_) return VILO_BAG_ERR_IO;
  struct stat st;
  if (fstat(fileno(f_), &st) != 0) { close(); return VILO_BAG_ERR_IO; }
  file_size_ = (long)st.st_size;
  char magic[MAGIC_N];
  if (std::fread(magic, 1, MAGIC_N, f_) != MAGIC_N || std::memcmp(magic, MAGIC, MAGIC_N) != 0) { close(); return VILO_BAG_ERR_FORMAT; }
  std::vector<uint8_t> h, d;
  std::map<std::string, std::string> hf;
  if (read_record(f_, file_size_, &h, &d) != 1 || !parse_fields(h.data(), h.size(), &hf) || field_op(hf) != 0x03 || !field_u64(hf, "index_pos", &index_pos) ||
      !field_u32(hf, "conn_count", &conn_count) || !field_u32(hf, "chunk_count", &chunk_count)) {
    close();
    return VILO_BAG_ERR_FORMAT;
  }
  chunk_.clear(); pos_ = 0; conns_.clear();
  return VILO_BAG_OK;
}
void BagReader::close() {
  if (f_) std::fclose(f_);
  f_ = nullptr;
}
int BagReader::load_next_chunk() {
  std::vector<uint8_t> h, d;
  std::map<std::string, std::string> hf;
  for (;;) {
    const int rc = read_record(f_, file_size_, &h, &d);
    if (rc == 0) return VILO_BAG_END;
    if (rc < 0) return rc;
    if (!parse_fields(h.data(), h.size(), &hf)) return VILO_BAG_ERR_FORMAT;
    const int op = field_op(hf);
    if (op == 0x05) {
      auto c = hf.find("compression");
      if (c == hf.end()) return VILO_BAG_ERR_FORMAT;
      uint32_t size;
      if (!field_u32(hf, "size", &size)) return VILO_BAG_ERR_FORMAT;
      if (c->second == "none") {
        if (size != d.size()) return VILO_BAG_ERR_FORMAT;
        chunk_.swap(d);
      } else {
        const int drc = chunk_decompress(c->second, d, size, &chunk_);
        if (drc != VILO_BAG_OK) return drc;
      }
      pos_ = 0;
      return VILO_BAG_OK;
    }
    if (op == 0x07) { if (!register_connection(hf, d, &conns_)) return VILO_BAG_ERR_FORMAT; continue; }
    if (op == 0x04 || op == 0x06 || op == 0x03) continue;   // index data, chunk info: not needed for a walk in file order
    return VILO_BAG_ERR_FORMAT;
  }
}
int BagReader::next(BagMessage *m) {
  if (!f_) return VILO_BAG_ERR_IO;
  std::map<std::string, std::string> hf;
  for (;;) {
    if (pos_ >= chunk_.size()) {
      const int rc = load_next_chunk();
      if (rc != VILO_BAG_OK) return rc;
      continue;
    }
    const uint8_t *p = chunk_.data();
    const size_t n = chunk_.size();
    if (n - pos_ < 4) return VILO_BAG_ERR_FORMAT;
    const uint32_t hl = get_u32(p + pos_);
    if (n - pos_ - 4 < (size_t)hl + 4) return VILO_BAG_ERR_FORMAT;
    const uint8_t *hp = p + pos_ + 4;
    const uint32_t dl = get_u32(hp + hl);
    if (n - pos_ - 8 - hl < dl) return VILO_BAG_ERR_FORMAT;
    const uint8_t *dp = hp + hl + 4;
    pos_ += 8 + (size_t)hl + dl;
    if (!parse_fields(hp, hl, &hf)) return VILO_BAG_ERR_FORMAT;
    const int op = field_op(hf);
    if (op == 0x07) {
      if (!register_connection(hf, std::vector<uint8_t>(dp, dp + dl), &conns_)) return VILO_BAG_ERR_FORMAT;
      continue;
    }
    if (op != 0x02) return VILO_BAG_ERR_FORMAT;
    uint32_t conn; uint64_t t;
    if (!field_u32(hf, "conn", &conn) || !field_u64(hf, "time", &t)) return VILO_BAG_ERR_FORMAT;
    auto c = conns_.find(conn);
    if (c == conns_.end()) return VILO_BAG_ERR_FORMAT;   // (a connection record precedes the connection's first message in every chunk)
    m->topic = c->second.first; m->type = c->second.second;
    m->secs = (uint32_t)(t & 0xffffffffULL); m->nsecs = (uint32_t)(t >> 32);
    m->data.assign(dp, dp + dl);
    return VILO_BAG_OK;
  }
}

}  // namespace vilo

// ---------------------------------------------------------------------------------------------------------------------------------
// C entry points
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
struct ReaderHandle {
  vilo::BagReader r;
  vilo::BagMessage m;
  vilo::ImageMsg img;
  vilo::PointCloudMsg pc;
  std::vector<float> channels;
};
void copy_str(char *dst, size_t cap, const std::string &s) {
  const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
  std::memcpy(dst, s.data(), n);
  dst[n] = 0;
}
}  // namespace

namespace {
// a C entry point's body: exceptions become the error code
template <class F>
int c_guard(int err, F f) {
  try {
    return f();
  } catch (...) {
    return err;
  }
}
}  // namespace

extern "C" {
void *vilo_bag_writer_open(const char *path, int chunk_threshold_bytes) {
  vilo::BagWriter *w = nullptr;
  try {
    w = new vilo::BagWriter;
    if (!w->open(path, chunk_threshold_bytes > 0 ? (size_t)chunk_threshold_bytes : 768 * 1024)) { delete w; return nullptr; }
    return w;
  } catch (...) {
    delete w;
    return nullptr;
  }
}
int vilo_bag_write_imu(void *h, const char *topic, uint32_t seq, uint32_t secs, uint32_t nsecs, const char *frame_id, const double *acc3, const double *gyr3) {
  return c_guard(-1, [&]() -> int {
    vilo::ImuMsg m;
    m.header.seq = seq; m.header.secs = secs; m.header.nsecs = nsecs; m.header.frame_id = frame_id ? frame_id : "";
    for (int i = 0; i < 3; ++i) { m.linear_acceleration[i] = acc3[i]; m.angular_velocity[i] = gyr3[i]; }
    std::vector<uint8_t> d;
    vilo::serialize(m, &d);
    return ((vilo::BagWriter *)h)->write(topic, vilo::BAG_IMU, secs, nsecs, d) ? 0 : -1;
  });
}
int vilo_bag_write_joint_state(void *h, const char *topic, uint32_t seq, uint32_t secs, uint32_t nsecs, int n, const double *position, const double *velocity,
                               const double *effort) {
  return c_guard(-1, [&]() -> int {
    vilo::JointStateMsg m;
    m.header.seq = seq; m.header.secs = secs; m.header.nsecs = nsecs;
    m.position.assign(position, position + n); m.velocity.assign(velocity, velocity + n); m.effort.assign(effort, effort + n);
    std::vector<uint8_t> d;
    vilo::serialize(m, &d);
    return ((vilo::BagWriter *)h)->write(topic, vilo::BAG_JOINT_STATE, secs, nsecs, d) ? 0 : -1;
  });
}
int vilo_bag_write_image(void *h, const char *topic, uint32_t seq, uint32_t secs, uint32_t nsecs, const char *frame_id, uint32_t height, uint32_t width,
                         const char *encoding, uint32_t step, const uint8_t *data) {
  return c_guard(-1, [&]() -> int {
    vilo::ImageMsg m;
    m.header.seq = seq; m.header.secs = secs; m.header.nsecs = nsecs; m.header.frame_id = frame_id ? frame_id : "";
    m.height = height; m.width = width; m.step = step; m.encoding = encoding ? encoding : "mono8";
    m.data.assign(data, data + (size_t)step * height);
    std::vector<uint8_t> d;
    vilo::serialize(m, &d);
    return ((vilo::BagWriter *)h)->write(topic, vilo::BAG_IMAGE, secs, nsecs, d) ? 0 : -1;
  });
}
int vilo_bag_write_point_cloud(void *h, const char *topic, uint32_t seq, uint32_t secs, uint32_t nsecs, int n_points, const float *xyz, int n_channels,
                               const char *const *channel_names, const float *channels) {
  return c_guard(-1, [&]() -> int {
    vilo::PointCloudMsg m;
    m.header.seq = seq; m.header.secs = secs; m.header.nsecs = nsecs;
    m.points.assign(xyz, xyz + 3 * (size_t)n_points);
    for (int c = 0; c < n_channels; ++c) {
      m.channel_name.push_back(channel_names[c]);
      m.channel_values.push_back(std::vector<float>(channels + (size_t)c * n_points, channels + (size_t)(c + 1) * n_points));
    }
    std::vector<uint8_t> d;
    vilo::serialize(m, &d);
    return ((vilo::BagWriter *)h)->write(topic, vilo::BAG_POINT_CLOUD, secs, nsecs, d) ? 0 : -1;
  });
}
int vilo_bag_writer_set_compression(void *h, const char *name) {
  return c_guard(vilo::VILO_BAG_ERR_FORMAT, [&]() -> int { return h && name ? ((vilo::BagWriter *)h)->setCompression(name) : vilo::VILO_BAG_ERR_FORMAT; });
}
int vilo_bag_writer_close(void *h) {
  vilo::BagWriter *w = (vilo::BagWriter *)h;
  const int rc = c_guard(-1, [&]() -> int { return w->close() ? 0 : -1; });
  delete w;
  return rc;
}

// (no C++ exception may cross the C boundary into ctypes / cgo: an allocation failure or a length error on a damaged bag is a return code)
void *vilo_bag_reader_open(const char *path, int *rc) {
  ReaderHandle *r = nullptr;
  try {
    r = new ReaderHandle;
    const int e = r->r.open(path);
    if (rc) *rc = e;
    if (e != vilo::VILO_BAG_OK) { delete r; return nullptr; }
    return r;
  } catch (...) {
    delete r;
    if (rc) *rc = vilo::VILO_BAG_ERR_FORMAT;
    return nullptr;
  }
}
int vilo_bag_reader_info(void *h, uint32_t *conn_count, uint32_t *chunk_count, uint64_t *index_pos) {
  ReaderHandle *r = (ReaderHandle *)h;
  *conn_count = r->r.conn_count; *chunk_count = r->r.chunk_count; *index_pos = r->r.index_pos;
  return 0;
}
static int bag_reader_next_impl(void *h, vilo_bag_msg *o);
int vilo_bag_reader_next(void *h, vilo_bag_msg *o) {
  try {
    return bag_reader_next_impl(h, o);
  } catch (const std::bad_alloc &) {
    return vilo::VILO_BAG_ERR_IO;
  } catch (...) {
    return vilo::VILO_BAG_ERR_FORMAT;
  }
}
static int bag_reader_next_impl(void *h, vilo_bag_msg *o) {
  ReaderHandle *r = (ReaderHandle *)h;
  const int rc = r->r.next(&r->m);
  if (rc != vilo::VILO_BAG_OK) return rc;
  std::memset(o, 0, sizeof(*o));
  o->kind = r->m.kind();
  o->rec_secs = r->m.secs; o->rec_nsecs = r->m.nsecs;
  copy_str(o->topic, sizeof(o->topic), r->m.topic);
  copy_str(o->type, sizeof(o->type), r->m.type);
  const uint8_t *p = r->m.data.data();
  const size_t n = r->m.data.size();
  auto set_header = [&](const vilo::RosHeader &hd) { o->seq = hd.seq; o->secs = hd.secs; o->nsecs = hd.nsecs; copy_str(o->frame_id, sizeof(o->frame_id), hd.frame_id); };
  if (o->kind == vilo::BAG_IMU) {
    vilo::ImuMsg m;
    if (!vilo::deserialize(p, n, &m)) return vilo::VILO_BAG_ERR_FORMAT;
    set_header(m.header);
    std::memcpy(o->orientation, m.orientation, sizeof(o->orientation));
    std::memcpy(o->angular_velocity, m.angular_velocity, sizeof(o->angular_velocity));
    std::memcpy(o->linear_acceleration, m.linear_acceleration, sizeof(o->linear_acceleration));
  } else if (o->kind == vilo::BAG_JOINT_STATE) {
    vilo::JointStateMsg m;
    if (!vilo::deserialize(p, n, &m)) return vilo::VILO_BAG_ERR_FORMAT;
    set_header(m.header);
    o->n_position = (int32_t)m.position.size(); o->n_velocity = (int32_t)m.velocity.size(); o->n_effort = (int32_t)m.effort.size();
    for (size_t i = 0; i < m.position.size() && i < 32; ++i) o->position[i] = m.position[i];
    for (size_t i = 0; i < m.velocity.size() && i < 32; ++i) o->velocity[i] = m.velocity[i];
    for (size_t i = 0; i < m.effort.size() && i < 32; ++i) o->effort[i] = m.effort[i];
  } else if (o->kind == vilo::BAG_IMAGE) {
    if (!vilo::deserialize(p, n, &r->img)) return vilo::VILO_BAG_ERR_FORMAT;
    set_header(r->img.header);
    o->height = r->img.height; o->width = r->img.width; o->step = r->img.step; o->is_bigendian = r->img.is_bigendian;
    copy_str(o->encoding, sizeof(o->encoding), r->img.encoding);
    o->data = r->img.data.data(); o->data_len = (uint32_t)r->img.data.size();
  } else if (o->kind == vilo::BAG_POINT_CLOUD) {
    if (!vilo::deserialize(p, n, &r->pc)) return vilo::VILO_BAG_ERR_FORMAT;
    set_header(r->pc.header);
    const size_t np = r->pc.points.size() / 3, nc = r->pc.channel_name.size();
    o->n_points = (int32_t)np; o->n_channels = (int32_t)nc;
    r->channels.assign(np * nc, 0.0f);
    for (size_t c = 0; c < nc; ++c) {
      if (c < 16) copy_str(o->channel_names[c], sizeof(o->channel_names[c]), r->pc.channel_name[c]);
      for (size_t i = 0; i < np && i < r->pc.channel_values[c].size(); ++i) r->channels[c * np + i] = r->pc.channel_values[c][i];
    }
    o->points = r->pc.points.data(); o->channels = r->channels.data();
  }
  return 0;
}
void vilo_bag_reader_close(void *h) { delete (ReaderHandle *)h; }
}
