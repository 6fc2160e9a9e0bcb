// pgo_replay.cpp -- ROS-free replay of the loop-closure node's input side: reads a recording of the
// two topics the ORORA node publishes (/orora/odom, /orora/cloud_local; written by `odometry --record`, rosmsg.h
// format = the ROS 1 wire bytes of nav_msgs/Odometry and sensor_msgs/PointCloud2) and does what process_pg does with
// them (laserPosegraphOptimization.cpp:417-492): pair the fronts of the two queues by stamp, pcl::fromROSMsg /
// getOdom, keyframe selection by accumulated translation (keyframe_meter_gap), VoxelGrid(0.4 m) downsample +
// makeAndSaveScancontextAndKeys -- on the MI355X through the SCManager shim -- and, per keyframe, what
// performSCLoopClosure does (PGO.cpp:556-571): detectLoopClosureID and the "Loop detected!" line.
// --verify-loops adds what process_icp does with every detected pair (PGO.cpp:589-612): doICPVirtualRelative -- submap
// assembly around the loop keyframe, VoxelGrid, ICP, the fitness gate -- on the keyframe clouds kept in HBM
// (rsx_kfstore / rsx_loop_verify), printing the reference's "[SC loop] ICP fitness test ..." lines (PGO.cpp:386-389).
// Without a pose-graph optimiser (GTSAM is not part of this image) keyframePosesUpdated stays the odometry pose of each
// keyframe, which is what the reference itself uses until the first iSAM2 update (PGO.cpp:489).
// --save-map <file.pcd> writes the cloud pubMap publishes (PGO.cpp:631-655: every 2nd keyframe through its pose,
// VoxelGrid of --map_viz_filter_size, default 0.4) as a binary PCD file: the map-saving utility of README.md:139.
//
// usage: pgo_replay <recording> [--keyframe_meter_gap 2.0] [--sc_dist_thres 0.45] [--exhaustive] [--devices 0,1,..]
//                   [--verify-loops] [--save-map map.pcd] [--map_viz_filter_size 0.4]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <iostream>
#include <string>
#include <vector>

#include "pcd.h"
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
    bool exhaustive = false, verify_loops = false;
    std::string map_path;
    float map_viz_filter_size = 0.4f;  // PGO.cpp:691 default of mapviz_filter_size
    std::vector<int> devices;
    for (int i = 2; i < argc; i++) {
      const std::string a = argv[i];
      if (a == "--keyframe_meter_gap" && i + 1 < argc) keyframe_meter_gap = std::atof(argv[++i]);
      else if (a == "--sc_dist_thres" && i + 1 < argc) sc_dist_thres = std::atof(argv[++i]);
      else if (a == "--exhaustive") exhaustive = true;
      else if (a == "--verify-loops") verify_loops = true;
      else if (a == "--save-map" && i + 1 < argc) map_path = argv[++i];
      else if (a == "--map_viz_filter_size" && i + 1 < argc) map_viz_filter_size = (float)std::atof(argv[++i]);
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
    // keyframeLaserClouds / keyframePosesUpdated (PGO.cpp:74-76), the clouds in HBM
    rsx_kfstore *kf = nullptr;
    std::vector<double> keyframePosesUpdated;  // 6 doubles per keyframe
    long n_accepted = 0;
    if (verify_loops || !map_path.empty()) {
      if (rsx_kfstore_create(devices.empty() ? 0 : devices[0], &kf) != RSX_OK) throw std::runtime_error(rsx_last_error_string());
      scManager.attachKeyframeStore(kf);
    }

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
        for (double v : {pose_curr.x, pose_curr.y, pose_curr.z, pose_curr.roll, pose_curr.pitch, pose_curr.yaw})
          keyframePosesUpdated.push_back(v);  // PGO.cpp:488-489
        n_keyframes++;
        if (n_keyframes < scManager.NUM_EXCLUDE_RECENT) continue;  // PGO.cpp:558
        auto r = scManager.detectLoopClosureID();                   // PGO.cpp:561
        if (r.first != -1) {
          std::printf("Loop detected! - between %d and %ld\n", r.first, n_keyframes - 1);  // PGO.cpp:566
          n_loops++;
          if (verify_loops) {  // process_icp -> doICPVirtualRelative(prev_node_idx, curr_node_idx) (PGO.cpp:596-599)
            rsx_loop_verify_result v;
            if (rsx_loop_verify(kf, r.first, (int32_t)(n_keyframes - 1), &keyframePosesUpdated[6 * (size_t)r.first], nullptr, &v) != RSX_OK)
              throw std::runtime_error(rsx_last_error_string());
            if (!v.accepted) {  // PGO.cpp:385-389, character for character (std::cout of a double / a float)
              std::cout << "[SC loop] ICP fitness test failed (" << v.fitness << " > " << 0.3f << "). Reject this SC loop." << std::endl;
            } else {
              std::cout << "[SC loop] ICP fitness test passed (" << v.fitness << " < " << 0.3f << "). Add this SC loop." << std::endl;
              n_accepted++;
            }
          }
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
    std::fflush(stdout);
    if (!map_path.empty()) {
      // pubMap (PGO.cpp:631-655): SKIP_FRAMES = 2, downSizeFilterMapPGO
      int64_t nk = 0, np = 0, m = 0;
      if (rsx_kfstore_size(kf, &nk, &np) != RSX_OK) throw std::runtime_error(rsx_last_error_string());
      std::vector<float> map((size_t)(np > 0 ? np : 1) * 4);
      if (rsx_kfstore_build_map(kf, keyframePosesUpdated.data(), (int64_t)(keyframePosesUpdated.size() / 6), 2, map_viz_filter_size, map.data(),
                                np > 0 ? np : 1, &m) != RSX_OK)
        throw std::runtime_error(rsx_last_error_string());
      pcd::write_binary_xyzi(map_path, map.data(), m);
      std::printf("map: %lld points from %lld keyframes -> %s\n", (long long)m, (long long)nk, map_path.c_str());
    }
    if (kf) rsx_kfstore_destroy(kf);
    std::printf("frames=%ld keyframes=%ld loops=%ld dropped_odom=%ld", n_frames, n_keyframes, n_loops, n_dropped);
    if (verify_loops) std::printf(" loops_accepted=%ld", n_accepted);
    std::printf("\n");
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "pgo_replay: %s\n", e.what());
    return 1;
  }
}
