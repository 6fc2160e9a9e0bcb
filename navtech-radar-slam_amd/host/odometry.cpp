// odometry.cpp -- file-based radar odometry entry, the counterpart of the upstream
// `outlier-robust-radar-odometry/src/odometry.cpp` that the reference launches through
// `$(find orora)/launch/run_orora.launch` with the args `seq_dir` and `do_slam`
// (/root/reference/launch/navtech_radar_slam_mulran.launch:5-8, README.md:27,54-60).
// The upstream file is NOT in the reference checkout (empty submodule); this entry keeps its
// observable surface -- it consumes `<seq_dir>/polar_oxford_form/*.png`, and produces the
// accumulated pose (/orora/odom) and the current scan's feature points (/orora/cloud_local,
// sc_pgo.launch:6-7) with identical stamps (PGO.cpp:417-436 pairs them by stamp) -- and calls
// the GPU hot path through the C-ABI:
//     rsx_cen2019_extract       polar image -> keypoints (+ Cartesian points)
//     rsx_orora_register_batch  matched points -> SE(2) motion
//     rsx_frontend_*            polar -> Cartesian image, ORB-style descriptors at the keypoints, brute-force Hamming
//                               knnMatch(2) + ratio test (what upstream does with cv::remap / cv::ORB / cv::BFMatcher)
// Upstream also prunes the matches with a PMC max-clique step before the solver; that stays out (host heuristic,
// SURVEY 8f): the ratio test + a cross check feed ORORA, whose GNC / consensus stages reject the remaining outliers.
// `--matcher nn` selects the round-1 stand-in instead (mutual nearest neighbours in the sensor frame, no descriptors).
//
// Output: one line per frame on stdout / --out file:  stamp_ns x y yaw n_keypoints n_matches
// With -DRSX_WITH_ROS (ROS 1 present) the same data is also published on /orora/odom and
// /orora/cloud_local; that part cannot be compiled in this container.
//
// Build: make -C navtech-radar-slam_amd/host odometry   (needs zlib for the PNG inflate)
#include <dirent.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "rosmsg.h"
#include "rsx.h"

#ifdef RSX_WITH_ROS
#include <nav_msgs/Odometry.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <sensor_msgs/point_cloud2_iterator.h>
#include <tf/transform_datatypes.h>
#endif

namespace {

constexpr int kMeta = 11;             // bytes in front of the power samples of each row
constexpr float kResolution = 0.0595f;  // Navtech CIR204-H range bin [m]

[[noreturn]] void die(const std::string &m) { throw std::runtime_error(m); }

void check(int status, const char *what) {
  if (status != RSX_OK) die(std::string(what) + ": rsx status " + std::to_string(status) + ": " + rsx_last_error_string());
}

uint32_t be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

// minimal PNG reader: 8-bit grayscale, non-interlaced (what MulRan's polar_oxford_form uses)
std::vector<uint8_t> read_png_gray8(const std::string &path, int *width, int *height) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) die("cannot open " + path);
  std::vector<uint8_t> buf;
  uint8_t tmp[65536];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  std::fclose(f);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (buf.size() < 8 || std::memcmp(buf.data(), sig, 8) != 0) die(path + ": not a PNG");
  size_t pos = 8;
  int w = 0, h = 0;
  std::vector<uint8_t> idat;
  while (pos + 12 <= buf.size()) {
    const uint32_t len = be32(&buf[pos]);
    const char *type = reinterpret_cast<const char *>(&buf[pos + 4]);
    if (pos + 12 + len > buf.size()) die(path + ": truncated chunk");
    const uint8_t *data = &buf[pos + 8];
    if (!std::memcmp(type, "IHDR", 4)) {
      w = (int)be32(data);
      h = (int)be32(data + 4);
      if (data[8] != 8 || data[9] != 0 || data[12] != 0) die(path + ": only 8-bit grayscale non-interlaced PNG is supported");
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + len;
  }
  if (w <= 0 || h <= 0) die(path + ": no IHDR");
  std::vector<uint8_t> raw((size_t)h * (w + 1));
  uLongf out_len = raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), idat.size()) != Z_OK || out_len != raw.size()) die(path + ": inflate failed");
  std::vector<uint8_t> img((size_t)h * w);
  for (int y = 0; y < h; y++) {  // undo the per-row filters (bpp = 1)
    const uint8_t ft = raw[(size_t)y * (w + 1)];
    const uint8_t *in = &raw[(size_t)y * (w + 1) + 1];
    uint8_t *out = &img[(size_t)y * w];
    const uint8_t *up = y ? &img[(size_t)(y - 1) * w] : nullptr;
    for (int x = 0; x < w; x++) {
      const int a = x ? out[x - 1] : 0, b = up ? up[x] : 0, c = (x && up) ? up[x - 1] : 0;
      int pred = 0;
      switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: {
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: die(path + ": bad PNG filter");
      }
      out[x] = (uint8_t)(in[x] + pred);
    }
  }
  *width = w;
  *height = h;
  return img;
}

struct Scan {
  int64_t stamp_ns = 0;
  std::vector<float> xy;       // keypoints in the sensor frame (x,y pairs)
  std::vector<uint8_t> desc;   // 32 bytes per keypoint (rsx_frontend_describe)
  std::vector<uint8_t> valid;  // descriptor available (patch inside the Cartesian image)
};

// mutual nearest neighbours within `gate` metres (stand-in for ORB + BF-Hamming + PMC)
void associate(const Scan &prev, const Scan &cur, float gate, std::vector<float> *src, std::vector<float> *dst) {
  const size_t np = prev.xy.size() / 2, nc = cur.xy.size() / 2;
  std::vector<int> fwd(np, -1), bwd(nc, -1);
  std::vector<float> bd(nc, gate * gate);
  for (size_t i = 0; i < np; i++) {
    float best = gate * gate;
    for (size_t j = 0; j < nc; j++) {
      const float dx = prev.xy[2 * i] - cur.xy[2 * j], dy = prev.xy[2 * i + 1] - cur.xy[2 * j + 1];
      const float d = dx * dx + dy * dy;
      if (d < best) {
        best = d;
        fwd[i] = (int)j;
      }
      if (d < bd[j]) {
        bd[j] = d;
        bwd[j] = (int)i;
      }
    }
  }
  for (size_t i = 0; i < np; i++)
    if (fwd[i] >= 0 && bwd[(size_t)fwd[i]] == (int)i) {
      src->push_back(prev.xy[2 * i]);
      src->push_back(prev.xy[2 * i + 1]);
      dst->push_back(cur.xy[2 * (size_t)fwd[i]]);
      dst->push_back(cur.xy[2 * (size_t)fwd[i] + 1]);
    }
}

}  // namespace

int main(int argc, char **argv) {
  try {
    std::string seq_dir, out_path, record_path, matcher = "orb";
    int max_frames = -1, device = 0;
    double rate_hz = 0.0;
    float gate = 6.0f;
    for (int i = 1; i < argc; i++) {
      const std::string a = argv[i];
      if (a == "--out" && i + 1 < argc) out_path = argv[++i];
      else if (a == "--record" && i + 1 < argc) record_path = argv[++i];  // ROS 1 wire bytes of both topics (rosmsg.h)
      else if (a == "--max_frames" && i + 1 < argc) max_frames = std::atoi(argv[++i]);
      else if (a == "--gate" && i + 1 < argc) gate = (float)std::atof(argv[++i]);
      else if (a == "--matcher" && i + 1 < argc) matcher = argv[++i];  // orb (default) | nn
      else if (a.rfind("seq_dir:=", 0) == 0) seq_dir = a.substr(9);  // roslaunch-style arg
      else if (a.rfind("do_slam:=", 0) == 0) continue;                // accepted for launch compatibility
      else if (a.rfind("device:=", 0) == 0) device = std::atoi(a.c_str() + 8);    // run_orora.launch
      else if (a.rfind("rate_hz:=", 0) == 0) rate_hz = std::atof(a.c_str() + 9);  // pacing of the publishers
      else if (a.rfind("__", 0) == 0 || a.find(":=") != std::string::npos) continue;  // roslaunch remaps (__name:=...)
      else seq_dir = a;
    }
    (void)rate_hz;  // only the ROS publishers are paced
    if (seq_dir.empty()) die("usage: odometry <seq_dir> [--out poses.txt] [--max_frames N] [--gate metres]");
    const std::string dir = seq_dir + "/polar_oxford_form";
    std::vector<std::string> files;
    if (DIR *d = opendir(dir.c_str())) {
      while (dirent *e = readdir(d)) {
        const std::string n = e->d_name;
        if (n.size() > 4 && n.substr(n.size() - 4) == ".png") files.push_back(n);
      }
      closedir(d);
    } else {
      die("cannot list " + dir);
    }
    std::sort(files.begin(), files.end());
    if (max_frames >= 0 && (int)files.size() > max_frames) files.resize((size_t)max_frames);
    if (files.empty()) die("no *.png under " + dir);

    FILE *out = out_path.empty() ? stdout : std::fopen(out_path.c_str(), "w");
    if (!out) die("cannot write " + out_path);
    FILE *rec = nullptr;
    if (!record_path.empty()) {
      rec = std::fopen(record_path.c_str(), "wb");
      if (!rec) die("cannot write " + record_path);
      std::fwrite(rosmsg::kReplayMagic, 1, 10, rec);
    }
    auto record = [&](uint8_t topic, const std::vector<uint8_t> &bytes) {
      const uint32_t len = (uint32_t)bytes.size();
      std::fwrite(&topic, 1, 1, rec);
      std::fwrite(&len, 4, 1, rec);
      std::fwrite(bytes.data(), 1, bytes.size(), rec);
    };
#ifdef RSX_WITH_ROS
    ros::init(argc, argv, "orora");
    ros::NodeHandle nh;
    ros::Publisher pub_odom = nh.advertise<nav_msgs::Odometry>("/orora/odom", 100);
    ros::Publisher pub_cloud = nh.advertise<sensor_msgs::PointCloud2>("/orora/cloud_local", 100);
#endif

    rsx_cen2019 *cen = nullptr;
    rsx_orora *reg = nullptr;
    rsx_frontend *fe = nullptr;
    const bool use_orb = matcher != "nn";
    if (matcher != "nn" && matcher != "orb") die("--matcher must be orb or nn");
    check(rsx_orora_create(device, &reg), "rsx_orora_create");
    rsx_cen2019_params cp;
    rsx_cen2019_default_params(&cp);
    double px = 0, py = 0, pyaw = 0;  // accumulated pose of the sensor in the odom frame
    Scan prev;
    int rows = 0, cols = 0;
    std::vector<int32_t> targets(2 * 200000);
    std::vector<float> xy(2 * 200000), az;
    for (size_t fi = 0; fi < files.size(); fi++) {
      int w = 0, h = 0;
      const std::vector<uint8_t> img = read_png_gray8(dir + "/" + files[fi], &w, &h);
      if (!cen) {
        rows = h;
        cols = w - kMeta;
        check(rsx_cen2019_create(device, rows, cols, &cen), "rsx_cen2019_create");
        if (use_orb) check(rsx_frontend_create(device, rows, cols, nullptr, &fe), "rsx_frontend_create");
        az.resize((size_t)rows);
      } else if (h != rows || w - kMeta != cols) {
        die(files[fi] + ": image shape changed");
      }
      Scan cur;
      std::memcpy(&cur.stamp_ns, &img[0], 8);  // little-endian int64 at bytes 0-7 of the first row
      if (cur.stamp_ns <= 0) cur.stamp_ns = std::atoll(files[fi].c_str());
      for (int a = 0; a < rows; a++) {
        uint16_t cnt;
        std::memcpy(&cnt, &img[(size_t)a * w + 8], 2);
        az[(size_t)a] = (float)((double)cnt * 2.0 * M_PI / 5600.0);
      }
      int32_t n = 0;
      check(rsx_cen2019_extract(cen, img.data(), w, kMeta, &cp, az.data(), kResolution, targets.data(), xy.data(), 200000, &n),
            "rsx_cen2019_extract");
      n = std::min(n, 200000);
      cur.xy.assign(xy.begin(), xy.begin() + 2 * (size_t)n);
      if (use_orb) {
        check(rsx_frontend_cartesian(fe, img.data(), w, kMeta, az.data(), kResolution, nullptr), "rsx_frontend_cartesian");
        cur.desc.resize((size_t)n * 32);
        cur.valid.resize((size_t)n);
        check(rsx_frontend_describe(fe, cur.xy.data(), n, cur.desc.data(), cur.valid.data()), "rsx_frontend_describe");
      }
      size_t n_match = 0;
      if (fi > 0) {
        std::vector<float> src, dst;
        if (use_orb) {
          // knnMatch(2) + ratio in both directions, kept when they agree (cross check)
          const int32_t np = (int32_t)(prev.xy.size() / 2);
          std::vector<int32_t> fwd((size_t)np, -1), bwd((size_t)n, -1);
          check(rsx_frontend_match(fe, prev.desc.data(), prev.valid.data(), np, cur.desc.data(), cur.valid.data(), n, 0.8f, fwd.data(),
                                   nullptr, nullptr), "rsx_frontend_match");
          check(rsx_frontend_match(fe, cur.desc.data(), cur.valid.data(), n, prev.desc.data(), prev.valid.data(), np, 0.8f, bwd.data(),
                                   nullptr, nullptr), "rsx_frontend_match");
          for (int32_t i = 0; i < np; i++)
            if (fwd[(size_t)i] >= 0 && bwd[(size_t)fwd[(size_t)i]] == i) {
              const size_t j = (size_t)fwd[(size_t)i];
              src.insert(src.end(), {prev.xy[2 * (size_t)i], prev.xy[2 * (size_t)i + 1]});
              dst.insert(dst.end(), {cur.xy[2 * j], cur.xy[2 * j + 1]});
            }
        } else {
          associate(prev, cur, gate, &src, &dst);
        }
        n_match = src.size() / 2;
        const size_t cap = (size_t)rsx_orora_max_correspondences();
        if (n_match > cap) {  // keep an evenly spread subset
          std::vector<float> s2, d2;
          for (size_t i = 0; i < cap; i++) {
            const size_t k = i * n_match / cap;
            s2.insert(s2.end(), {src[2 * k], src[2 * k + 1]});
            d2.insert(d2.end(), {dst[2 * k], dst[2 * k + 1]});
          }
          src.swap(s2);
          dst.swap(d2);
          n_match = cap;
        }
        const int64_t offsets[2] = {0, (int64_t)n_match};
        rsx_orora_result r;
        // src = current scan, dst = previous scan: the motion of the sensor expressed in the previous frame
        check(rsx_orora_register_batch(reg, dst.data(), src.data(), offsets, 1, nullptr, &r), "rsx_orora_register_batch");
        if (r.status == 0) {
          const double c = std::cos(pyaw), s = std::sin(pyaw);
          px += c * r.x - s * r.y;
          py += s * r.x + c * r.y;
          pyaw += r.yaw;
        }
      }
      std::fprintf(out, "%lld %.6f %.6f %.6f %d %zu\n", (long long)cur.stamp_ns, px, py, pyaw, n, n_match);
      if (rec) {  // what the publishers below put on /orora/odom and /orora/cloud_local, as ROS 1 wire bytes
        rosmsg::Header h;
        h.seq = (uint32_t)fi;
        h.fromNSec(cur.stamp_ns);
        h.frame_id = "odom";
        const double pos[3] = {px, py, 0.0}, quat[4] = {0.0, 0.0, std::sin(0.5 * pyaw), std::cos(0.5 * pyaw)};
        record(rosmsg::kOdom, rosmsg::serialize_odometry(h, "radar", pos, quat));
        h.frame_id = "radar";
        std::vector<rosmsg::PointXYZI> pc((size_t)n);
        for (int k = 0; k < n; k++) pc[(size_t)k] = rosmsg::PointXYZI{cur.xy[2 * (size_t)k], cur.xy[2 * (size_t)k + 1], 0.f, 0.f};
        record(rosmsg::kCloud, rosmsg::serialize_pointcloud2(h, pc));
      }
#ifdef RSX_WITH_ROS
      ros::Time stamp;
      stamp.fromNSec((uint64_t)cur.stamp_ns);
      nav_msgs::Odometry od;
      od.header.stamp = stamp;
      od.header.frame_id = "odom";
      od.pose.pose.position.x = px;
      od.pose.pose.position.y = py;
      od.pose.pose.orientation = tf::createQuaternionMsgFromYaw(pyaw);
      pub_odom.publish(od);
      sensor_msgs::PointCloud2 pc;
      pc.header = od.header;
      pc.header.frame_id = "radar";
      sensor_msgs::PointCloud2Modifier mod(pc);
      mod.setPointCloud2Fields(4, "x", 1, sensor_msgs::PointField::FLOAT32, "y", 1, sensor_msgs::PointField::FLOAT32, "z", 1,
                               sensor_msgs::PointField::FLOAT32, "intensity", 1, sensor_msgs::PointField::FLOAT32);
      mod.resize((size_t)n);
      sensor_msgs::PointCloud2Iterator<float> ix(pc, "x"), iy(pc, "y"), iz(pc, "z"), ii(pc, "intensity");
      for (int k = 0; k < n; k++, ++ix, ++iy, ++iz, ++ii) {
        *ix = cur.xy[2 * (size_t)k];
        *iy = cur.xy[2 * (size_t)k + 1];
        *iz = 0.f;
        *ii = 0.f;
      }
      pub_cloud.publish(pc);
      ros::spinOnce();
      if (rate_hz > 0.0) ros::Duration(1.0 / rate_hz).sleep();
      if (!ros::ok()) break;
#endif
      prev = std::move(cur);
    }
    if (out != stdout) std::fclose(out);
    if (rec) std::fclose(rec);
    rsx_cen2019_destroy(cen);
    rsx_orora_destroy(reg);
    rsx_frontend_destroy(fe);
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "odometry: %s\n", e.what());
    return 1;
  }
}
