// rosmsg.h -- ROS 1 wire format of the two messages that cross the boundary between the ORORA node and the
// loop-closure node: sensor_msgs/PointCloud2 on /orora/cloud_local and nav_msgs/Odometry on
// /orora/odom (sc_pgo.launch:6-7).  laserPosegraphOptimization.cpp turns them into a pcl::PointCloud<PointXYZI>
// with pcl::fromROSMsg (PGO.cpp:431-432) and into x, y, z, roll, pitch, yaw with getOdom (PGO.cpp:175-187); the
// functions below do the same on the serialised bytes, so that a recorded run can be replayed through the GPU
// path without ROS (host/pgo_replay.cpp).  Host glue only, header-only, little-endian hosts.
//
// Layout (ROS 1 serialisation: fields in declaration order, little-endian, string / array = uint32 length + bytes):
//   std_msgs/Header          uint32 seq; uint32 stamp.sec; uint32 stamp.nsec; string frame_id
//   sensor_msgs/PointCloud2  Header; uint32 height; uint32 width; PointField[] fields {string name; uint32 offset;
//                            uint8 datatype; uint32 count}; uint8 is_bigendian; uint32 point_step; uint32 row_step;
//                            uint8[] data; uint8 is_dense
//   nav_msgs/Odometry        Header; string child_frame_id; float64 position[3]; float64 orientation[4] (x, y, z, w);
//                            float64 pose covariance[36]; float64 linear[3]; float64 angular[3]; float64 twist covariance[36]
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rosmsg {

struct Header {
  uint32_t seq = 0;
  uint32_t sec = 0, nsec = 0;
  std::string frame_id;
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }  // ros::Time::toSec
  int64_t toNSec() const { return (int64_t)sec * 1000000000ll + (int64_t)nsec; }
  void fromNSec(int64_t t) {
    sec = (uint32_t)(t / 1000000000ll);
    nsec = (uint32_t)(t % 1000000000ll);
  }
};

struct PointXYZI {  // the payload of pcl::PointXYZI that the PGO node reads
  float x, y, z, intensity;
};

struct Pose6D {  // PGO.cpp:48-55
  double x, y, z, roll, pitch, yaw;
};

// ---- byte stream helpers ----
class Writer {
 public:
  std::vector<uint8_t> buf;
  template <typename T>
  void put(T v) {
    const size_t o = buf.size();
    buf.resize(o + sizeof(T));
    std::memcpy(&buf[o], &v, sizeof(T));
  }
  void str(const std::string &s) {
    put<uint32_t>((uint32_t)s.size());
    buf.insert(buf.end(), s.begin(), s.end());
  }
  void header(const Header &h) {
    put<uint32_t>(h.seq);
    put<uint32_t>(h.sec);
    put<uint32_t>(h.nsec);
    str(h.frame_id);
  }
};

class Reader {
 public:
  Reader(const uint8_t *p, size_t n) : p_(p), n_(n) {}
  template <typename T>
  T get() {
    need(sizeof(T));
    T v;
    std::memcpy(&v, p_ + o_, sizeof(T));
    o_ += sizeof(T);
    return v;
  }
  std::string str() {
    const uint32_t n = get<uint32_t>();
    need(n);
    std::string s(reinterpret_cast<const char *>(p_ + o_), n);
    o_ += n;
    return s;
  }
  const uint8_t *bytes(size_t n) {
    need(n);
    const uint8_t *r = p_ + o_;
    o_ += n;
    return r;
  }
  Header header() {
    Header h;
    h.seq = get<uint32_t>();
    h.sec = get<uint32_t>();
    h.nsec = get<uint32_t>();
    h.frame_id = str();
    return h;
  }
  size_t offset() const { return o_; }

 private:
  void need(size_t n) const {
    if (o_ + n > n_) throw std::runtime_error("rosmsg: truncated message");
  }
  const uint8_t *p_;
  size_t n_, o_ = 0;
};

// ---- sensor_msgs/PointCloud2 ----
constexpr uint8_t kFloat32 = 7;  // sensor_msgs/PointField FLOAT32

// what pcl::toROSMsg produces for a PointXYZI cloud: fields x, y, z at 0, 4, 8 and intensity at 16, 32-byte points
inline std::vector<uint8_t> serialize_pointcloud2(const Header &h, const std::vector<PointXYZI> &pts) {
  Writer w;
  w.header(h);
  w.put<uint32_t>(1);                     // height
  w.put<uint32_t>((uint32_t)pts.size());  // width
  w.put<uint32_t>(4);                     // fields
  const char *names[4] = {"x", "y", "z", "intensity"};
  const uint32_t offs[4] = {0, 4, 8, 16};
  for (int i = 0; i < 4; i++) {
    w.str(names[i]);
    w.put<uint32_t>(offs[i]);
    w.put<uint8_t>(kFloat32);
    w.put<uint32_t>(1);
  }
  w.put<uint8_t>(0);   // is_bigendian
  w.put<uint32_t>(32);  // point_step
  w.put<uint32_t>((uint32_t)(32 * pts.size()));
  w.put<uint32_t>((uint32_t)(32 * pts.size()));
  for (const PointXYZI &p : pts) {
    const float rec[8] = {p.x, p.y, p.z, 1.0f, p.intensity, 0.f, 0.f, 0.f};
    const size_t o = w.buf.size();
    w.buf.resize(o + 32);
    std::memcpy(&w.buf[o], rec, 32);
  }
  w.put<uint8_t>(1);  // is_dense
  return w.buf;
}

// pcl::fromROSMsg into PointXYZI: fields are matched BY NAME (any offsets / point_step / extra fields); a missing
// intensity field gives 0.  Only FLOAT32 little-endian fields are accepted (what the ORORA node publishes).
inline Header deserialize_pointcloud2(const uint8_t *data, size_t n, std::vector<PointXYZI> *out) {
  Reader r(data, n);
  const Header h = r.header();
  const uint32_t height = r.get<uint32_t>(), width = r.get<uint32_t>();
  const uint32_t nf = r.get<uint32_t>();
  int off[4] = {-1, -1, -1, -1};
  for (uint32_t i = 0; i < nf; i++) {
    const std::string name = r.str();
    const uint32_t o = r.get<uint32_t>();
    const uint8_t type = r.get<uint8_t>();
    (void)r.get<uint32_t>();
    const int k = name == "x" ? 0 : name == "y" ? 1 : name == "z" ? 2 : name == "intensity" ? 3 : -1;
    if (k >= 0) {
      if (type != kFloat32) throw std::runtime_error("rosmsg: field " + name + " is not FLOAT32");
      off[k] = (int)o;
    }
  }
  if (r.get<uint8_t>() != 0) throw std::runtime_error("rosmsg: big-endian point data");
  const uint32_t point_step = r.get<uint32_t>(), row_step = r.get<uint32_t>();
  const uint32_t nbytes = r.get<uint32_t>();
  const uint8_t *pd = r.bytes(nbytes);
  (void)r.get<uint8_t>();  // is_dense
  if (off[0] < 0 || off[1] < 0 || off[2] < 0) throw std::runtime_error("rosmsg: cloud without x / y / z");
  // everything below indexes the data buffer with numbers that came out of the message: check them first
  for (int k = 0; k < 4; k++)
    if (off[k] >= 0 && (uint64_t)off[k] + 4 > (uint64_t)point_step) throw std::runtime_error("rosmsg: field offset beyond point_step");
  if (point_step == 0 || (uint64_t)row_step < (uint64_t)width * point_step) throw std::runtime_error("rosmsg: bad point_step / row_step");
  if ((uint64_t)height * width > (uint64_t)nbytes / point_step) throw std::runtime_error("rosmsg: point data shorter than height x width");
  out->clear();
  out->reserve((size_t)height * width);
  for (uint32_t row = 0; row < height; row++)
    for (uint32_t col = 0; col < width; col++) {
      const size_t base = (size_t)row * row_step + (size_t)col * point_step;
      if (base + point_step > nbytes) throw std::runtime_error("rosmsg: point data shorter than height x width");
      PointXYZI p{0, 0, 0, 0};
      std::memcpy(&p.x, pd + base + off[0], 4);
      std::memcpy(&p.y, pd + base + off[1], 4);
      std::memcpy(&p.z, pd + base + off[2], 4);
      if (off[3] >= 0) std::memcpy(&p.intensity, pd + base + off[3], 4);
      out->push_back(p);
    }
  return h;
}

// ---- nav_msgs/Odometry ----
inline std::vector<uint8_t> serialize_odometry(const Header &h, const std::string &child_frame, const double pos[3],
                                               const double quat_xyzw[4]) {
  Writer w;
  w.header(h);
  w.str(child_frame);
  for (int i = 0; i < 3; i++) w.put<double>(pos[i]);
  for (int i = 0; i < 4; i++) w.put<double>(quat_xyzw[i]);
  for (int i = 0; i < 36; i++) w.put<double>(0.0);
  for (int i = 0; i < 6; i++) w.put<double>(0.0);
  for (int i = 0; i < 36; i++) w.put<double>(0.0);
  return w.buf;
}

// tf::Matrix3x3(tf::Quaternion(x, y, z, w)).getRPY(roll, pitch, yaw), as getOdom uses it (PGO.cpp:181-183)
inline void quaternion_to_rpy(const double q[4], double *roll, double *pitch, double *yaw) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double d = x * x + y * y + z * z + w * w;
  const double s = 2.0 / d;
  const double xs = x * s, ys = y * s, zs = z * s;
  const double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
  const double m01 = xy - wz, m02 = xz + wy;
  if (std::fabs(m20) >= 1.0) {  // gimbal lock branch of getEulerYPR
    *yaw = 0.0;
    const double delta = std::atan2(m01, m02);
    if (m20 < 0) {
      *pitch = M_PI / 2.0;
      *roll = delta;
    } else {
      *pitch = -M_PI / 2.0;
      *roll = delta;
    }
    return;
  }
  *pitch = -std::asin(m20);
  const double c = std::cos(*pitch);
  *roll = std::atan2(m21 / c, m22 / c);
  *yaw = std::atan2(m10 / c, m00 / c);
}

inline Header deserialize_odometry(const uint8_t *data, size_t n, Pose6D *pose, std::string *child_frame = nullptr) {
  Reader r(data, n);
  const Header h = r.header();
  const std::string child = r.str();
  if (child_frame) *child_frame = child;
  double pos[3], q[4];
  for (int i = 0; i < 3; i++) pos[i] = r.get<double>();
  for (int i = 0; i < 4; i++) q[i] = r.get<double>();
  (void)r.bytes(36 * 8 + 6 * 8 + 36 * 8);
  pose->x = pos[0];
  pose->y = pos[1];
  pose->z = pos[2];
  quaternion_to_rpy(q, &pose->roll, &pose->pitch, &pose->yaw);
  return h;
}

// ---- replay file: what a rosbag of the two topics boils down to.  "RSXREPLAY1" then records
//      {uint8 topic (0 = /orora/odom, 1 = /orora/cloud_local), uint32 length, serialised message} ----
constexpr char kReplayMagic[11] = "RSXREPLAY1";
enum Topic : uint8_t { kOdom = 0, kCloud = 1 };

}  // namespace rosmsg
