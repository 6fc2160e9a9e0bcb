// pgo_replay.cpp -- ROS-free replay of the loop-closure node's input side: reads a recording of the
// two topics the ORORA node publishes (/orora/odom, /orora/cloud_local; written by `odometry --record`, rosmsg.h
// format = the ROS 1 wire bytes of nav_msgs/Odometry and sensor_msgs/PointCloud2) and does what process_pg does with
// them (laserPosegraphOptimization.cpp:417-492): pair the fronts of the two queues by stamp, pcl::fromROSMsg /
// getOdom, keyframe selection by accumulated translation (keyframe_meter_gap), VoxelGrid(0.4 m) downsample +
// makeAndSaveScancontextAndKeys -- on the MI355X through the SCManager shim -- and, per keyframe, what
// performSCLoopClosure does (PGO.cpp:556-571): detectLoopClosureID and the "Loop detected!" line.
// The pose graph itself (GTSAM / ICP) is out of scope; this is the ScanContext side of the node, replayable offline.
//
// usage: pgo_replay <recording> [--keyframe_meter_gap 2.0] [--sc_dist_thres 0.45] [--exhaustive] [--devices 0,1,..]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <string>
#include <vector>

#include "rosmsg.h"
#include "scancontext/Scancontext.h"

struct Pt32 {  // pcl::PointXYZI memory layout
  float x, y, z, pad0, intensity, pad1[3];
};

int main(int argc, char **argv) {
  try {
    if (argc < 2) {
      std::fprintf(stderr, "usage: pgo_replay <recording> [--keyframe_meter_gap m] [--sc_dist_thres d] [--exhaustive] [--devices a,b]\n");
      return 1;
    }
    double keyframe_meter_gap = 2.0, sc_dist_thres = 0.2;  // PGO.cpp:676-677 defaults (sc_pgo.launch sets 0.2 / 0.45)
    bool exhaustive = false;
    std::vector<int> devices;
    for (int i = 2; i < argc; i++) {
      const std::string a = argv[i];
      if (a == "--keyframe_meter_gap" && i + 1 < argc) keyframe_meter_gap = std::atof(argv[++i]);
      else if (a == "--sc_dist_thres" && i + 1 < argc) sc_dist_thres = std::atof(argv[++i]);
      else if (a == "--exhaustive") exhaustive = true;
      else if (a == "--devices" && i + 1 < argc)
        for (const char *p = argv[++i]; *p;) {
          devices.push_back(std::atoi(p));
          while (*p && *p != ',') p++;
          if (*p == ',') p++;
        }
    }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + argv[1]);
    char magic[10];
    if (std::fread(magic, 1, 10, f) != 10 || std::memcmp(magic, rosmsg::kReplayMagic, 10) != 0)
      throw std::runtime_error("not an RSXREPLAY1 recording");

    SCManager scManager;  // PGO.cpp:99
    scManager.setSCdistThres(sc_dist_thres);  // PGO.cpp:685
    if (devices.size() > 1) scManager.setDevices(devices);
    else if (devices.size() == 1) scManager.setDevice(devices[0]);
    scManager.setExhaustive(exhaustive);
    scManager.setVerbose(false);

    // laserOdometryHandler / laserCloudFullResHandler: the two queues (PGO.cpp:124-137)
    struct Msg {
      rosmsg::Header h;
      std::vector<uint8_t> bytes;
    };
    std::deque<Msg> odometryBuf, fullResBuf;
    rosmsg::Pose6D odom_pose_prev{0, 0, 0, 0, 0, 0}, odom_pose_curr{0, 0, 0, 0, 0, 0};
    double movementAccumulation = 1000000.0;  // PGO.cpp:58: large value so that the first frame is a keyframe
    long n_frames = 0, n_keyframes = 0, n_loops = 0, n_dropped = 0;
    std::vector<rosmsg::PointXYZI> cloud;
    std::vector<Pt32> pts;

    auto process = [&]() {  // the body of process_pg's while loop (PGO.cpp:419-492), minus GPS and the pose graph
      while (!odometryBuf.empty() && !fullResBuf.empty()) {
        while (!odometryBuf.empty() && odometryBuf.front().h.toSec() < fullResBuf.front().h.toSec()) {
          odometryBuf.pop_front();  // PGO.cpp:425-426: odometry older than the cloud is dropped
          n_dropped++;
        }
        if (odometryBuf.empty()) break;
        rosmsg::deserialize_pointcloud2(fullResBuf.front().bytes.data(), fullResBuf.front().bytes.size(), &cloud);  // pcl::fromROSMsg
        fullResBuf.pop_front();
        rosmsg::Pose6D pose_curr;
        rosmsg::deserialize_odometry(odometryBuf.front().bytes.data(), odometryBuf.front().bytes.size(), &pose_curr);  // getOdom
        odometryBuf.pop_front();
        n_frames++;
        odom_pose_prev = odom_pose_curr;
        odom_pose_curr = pose_curr;
        const double dx = odom_pose_prev.x - odom_pose_curr.x, dy = odom_pose_prev.y - odom_pose_curr.y,
                     dz = odom_pose_prev.z - odom_pose_curr.z;
        movementAccumulation += std::sqrt(dx * dx + dy * dy + dz * dz);  // transDiff, PGO.cpp:188-191,459-460
        if (!(movementAccumulation > keyframe_meter_gap)) continue;     // PGO.cpp:462-470
        movementAccumulation = 0.0;
        pts.clear();
        for (const rosmsg::PointXYZI &p : cloud) pts.push_back(Pt32{p.x, p.y, p.z, 1.0f, p.intensity, {0, 0, 0}});
        // downSizeFilterScancontext.filter + scManager.makeAndSaveScancontextAndKeys (PGO.cpp:482-492) on the GPU, for one
        // device (fused) and for several (downsample on the first, same cloud to every shard) alike
        scManager.makeAndSaveScancontextAndKeysDownsampled(pts.empty() ? nullptr : &pts[0].x, pts.size(), sizeof(Pt32), 0.4f);
        n_keyframes++;
        if (n_keyframes < scManager.NUM_EXCLUDE_RECENT) continue;  // PGO.cpp:558
        auto r = scManager.detectLoopClosureID();                   // PGO.cpp:561
        if (r.first != -1) {
          std::printf("Loop detected! - between %d and %ld\n", r.first, n_keyframes - 1);  // PGO.cpp:566
          n_loops++;
        }
      }
    };

    for (;;) {
      uint8_t topic;
      uint32_t len;
      if (std::fread(&topic, 1, 1, f) != 1) break;
      if (std::fread(&len, 4, 1, f) != 1) throw std::runtime_error("truncated recording");
      Msg m;
      m.bytes.resize(len);
      if (len && std::fread(m.bytes.data(), 1, len, f) != len) throw std::runtime_error("truncated recording");
      rosmsg::Reader r(m.bytes.data(), m.bytes.size());
      m.h = r.header();
      if (topic == rosmsg::kOdom) odometryBuf.push_back(std::move(m));
      else if (topic == rosmsg::kCloud) fullResBuf.push_back(std::move(m));
      process();
    }
    std::fclose(f);
    std::printf("frames=%ld keyframes=%ld loops=%ld dropped_odom=%ld\n", n_frames, n_keyframes, n_loops, n_dropped);
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "pgo_replay: %s\n", e.what());
    return 1;
  }
}
