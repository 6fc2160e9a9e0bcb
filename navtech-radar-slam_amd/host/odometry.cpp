// odometry.cpp -- file-based radar odometry entry, the counterpart of the upstream
// `outlier-robust-radar-odometry/src/odometry.cpp` that the reference launches through
// `$(find orora)/launch/run_orora.launch` with the args `seq_dir` and `do_slam`
// (/root/reference/launch/navtech_radar_slam_mulran.launch:5-8, README.md:27,54-60).
// The upstream file is NOT in the reference checkout (empty submodule); this entry keeps its
// observable surface -- it consumes `<seq_dir>/polar_oxford_form/*.png`, and produces the
// accumulated pose (/orora/odom) and the current scan's feature points (/orora/cloud_local,
// sc_pgo.launch:6-7) with identical stamps (PGO.cpp:417-436 pairs them by stamp) -- and calls
// the GPU hot path through the C-ABI.  Default (round 3): the sequence is on disk, so it is processed in WINDOWS --
//     rsx_odometry_push         W decoded scans -> cen2019 keypoints, Cartesian images, ORB-style descriptors, knnMatch(2) +
//                               ratio + cross check of every consecutive pair, ORORA for all pairs of the window in ONE
//                               batch; everything between the image upload and 48 bytes per scan stays on the GPU
// while a pool of host threads inflates the PNGs of the next window into page-locked memory.  `--per-scan` keeps the
// round-2 loop (one scan at a time through the single-call entries, every intermediate through host vectors):
//     rsx_cen2019_extract       polar image -> keypoints (+ Cartesian points)
//     rsx_orora_register_batch  matched points -> SE(2) motion
//     rsx_frontend_*            polar -> Cartesian image, ORB-style descriptors at the keypoints, brute-force Hamming
//                               knnMatch(2) + ratio test (what upstream does with cv::remap / cv::ORB / cv::BFMatcher)
// Between matcher and solver the matches are pruned to the max clique of their distance-consistency graph, as upstream does
// with the PMC library (round 6: csrc/pmc.hip behind RSX_ORORA_PMC, on by default; `--no-pmc` feeds the solver every
// cross-checked ratio match, the pipeline of rounds 3-5).
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
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "png_gray8.h"
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

// host cores this process may use: hardware threads, cut by the cgroup v2 CPU quota when there is one
int usable_cores() {
  int n = (int)std::max(1u, std::thread::hardware_concurrency());
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32] = {0};
    long long period = 0;
    if (std::fscanf(f, "%31s %lld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
      const long long q = std::atoll(quota);
      if (q > 0) n = (int)std::min<long long>(n, (q + period - 1) / period);
    }
    std::fclose(f);
  }
  return std::max(1, n);
}

void check(int status, const char *what) {
  if (status != RSX_OK) die(std::string(what) + ": rsx status " + std::to_string(status) + ": " + rsx_last_error_string());
}

using rsxhost::read_png_gray8;  // png_gray8.h

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
    int max_frames = -1, device = 0, window = 0, threads = 0;
    bool per_scan = false, timing = false, use_pmc = true;
    double rate_hz = 0.0;
    float gate = 6.0f;
    for (int i = 1; i < argc; i++) {
      const std::string a = argv[i];
      if (a == "--out" && i + 1 < argc) out_path = argv[++i];
      else if (a == "--record" && i + 1 < argc) record_path = argv[++i];  // ROS 1 wire bytes of both topics (rosmsg.h)
      else if (a == "--max_frames" && i + 1 < argc) max_frames = std::atoi(argv[++i]);
      else if (a == "--gate" && i + 1 < argc) gate = (float)std::atof(argv[++i]);
      else if (a == "--matcher" && i + 1 < argc) matcher = argv[++i];  // orb (default) | nn
      else if (a == "--window" && i + 1 < argc) window = std::atoi(argv[++i]);    // scans per rsx_odometry_push (default: two of the library's windows)
      else if (a == "--threads" && i + 1 < argc) threads = std::atoi(argv[++i]);  // PNG decode threads (default: twice the usable cores, <= 64)
      else if (a == "--no-pmc") use_pmc = false;                                   // skip the max-clique inlier selection before the solver
      else if (a == "--per-scan") per_scan = true;                                // the round-2 loop: one scan per call, host vectors in between
      else if (a == "--timing") timing = true;                                    // decode / pipeline seconds on stderr
      else if (a.rfind("seq_dir:=", 0) == 0) seq_dir = a.substr(9);  // roslaunch-style arg
      else if (a.rfind("do_slam:=", 0) == 0) continue;                // accepted for launch compatibility
      else if (a.rfind("device:=", 0) == 0) device = std::atoi(a.c_str() + 8);    // run_orora.launch
      else if (a.rfind("rate_hz:=", 0) == 0) rate_hz = std::atof(a.c_str() + 9);  // pacing of the publishers
      else if (a.rfind("__", 0) == 0 || a.find(":=") != std::string::npos) continue;  // roslaunch remaps (__name:=...)
      else seq_dir = a;
    }
    (void)rate_hz;  // only the ROS publishers are paced
    if (seq_dir.empty())
      die("usage: odometry <seq_dir> [--out poses.txt] [--max_frames N] [--matcher orb|nn] [--window W] [--threads T] [--per-scan] [--no-pmc] [--timing]");
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

    double px = 0, py = 0, pyaw = 0;  // accumulated pose of the sensor in the odom frame
    // one line per frame, the recording and the publishers: shared by both loops
    auto emit = [&](size_t fi, int64_t stamp_ns, int n, size_t n_match, const float *pts) {
      std::fprintf(out, "%lld %.6f %.6f %.6f %d %zu\n", (long long)stamp_ns, px, py, pyaw, n, n_match);
      if (rec) {  // what the publishers below put on /orora/odom and /orora/cloud_local, as ROS 1 wire bytes
        rosmsg::Header h;
        h.seq = (uint32_t)fi;
        h.fromNSec(stamp_ns);
        h.frame_id = "odom";
        const double pos[3] = {px, py, 0.0}, quat[4] = {0.0, 0.0, std::sin(0.5 * pyaw), std::cos(0.5 * pyaw)};
        record(rosmsg::kOdom, rosmsg::serialize_odometry(h, "radar", pos, quat));
        h.frame_id = "radar";
        std::vector<rosmsg::PointXYZI> pc((size_t)n);
        for (int k = 0; k < n; k++) pc[(size_t)k] = rosmsg::PointXYZI{pts[2 * (size_t)k], pts[2 * (size_t)k + 1], 0.f, 0.f};
        record(rosmsg::kCloud, rosmsg::serialize_pointcloud2(h, pc));
      }
#ifdef RSX_WITH_ROS
      ros::Time stamp;
      stamp.fromNSec((uint64_t)stamp_ns);
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
        *ix = pts[2 * (size_t)k];
        *iy = pts[2 * (size_t)k + 1];
        *iz = 0.f;
        *ii = 0.f;
      }
      pub_cloud.publish(pc);
      ros::spinOnce();
      if (rate_hz > 0.0) ros::Duration(1.0 / rate_hz).sleep();
#endif
    };
    auto compose = [&](const rsx_orora_result &r) {
      if (r.status != 0) return;
      const double c = std::cos(pyaw), s = std::sin(pyaw);
      px += c * r.x - s * r.y;
      py += s * r.x + c * r.y;
      pyaw += r.yaw;
    };
    if (matcher != "nn" && matcher != "orb") die("--matcher must be orb or nn");

    if (matcher == "orb" && !per_scan) {
      // ---------------- windows of scans through rsx_odometry_push ----------------
      using clk = std::chrono::steady_clock;
      int w0 = 0, h0 = 0;
      (void)read_png_gray8(dir + "/" + files[0], &w0, &h0);
      const int rows = h0, cols = w0 - kMeta;
      if (cols < 2) die(files[0] + ": image too narrow for " + std::to_string(kMeta) + " metadata bytes per row");
      rsx_odometry_params op;
      check(rsx_odometry_default_params(&op), "rsx_odometry_default_params");
      op.device = device;
      op.radar_resolution = kResolution;
      op.col_offset = kMeta;
      if (!use_pmc) op.orora.flags &= ~RSX_ORORA_PMC;
      rsx_odometry *odo = nullptr;
      check(rsx_odometry_create(&op, rows, cols, &odo), "rsx_odometry_create");
      // scans per rsx_odometry_push: two of the library's internal windows, so that inside a call the upload and the extraction of
      // the second overlap the matching of the first (the pinned buffers are 2 x W images)
      const int W = window > 0 ? std::min(window, 4096) : 2 * rsx_odometry_window();
      // decode threads: twice the cores this process may use -- the visible CPUs cut by the cgroup's CPU quota (a container
      // that shows 256 CPUs under a quota of 16 throttles 64 busy threads: 2.1-2.6 k scans/s where 32 threads reach 3.6-3.8 k)
      const int T = threads > 0 ? threads : std::max(1, std::min(64, 2 * usable_cores()));
      const size_t ibytes = (size_t)rows * w0;
      struct Win {
        uint8_t *img = nullptr;
        std::vector<float> az;
        std::vector<int64_t> stamp;
        int n = 0;
        double decode_cpu_s = 0;
      } wins[2];
      for (Win &wn : wins) {
        void *p = nullptr;
        check(rsx_host_alloc_pinned(ibytes * (size_t)W, &p), "rsx_host_alloc_pinned");
        wn.img = static_cast<uint8_t *>(p);
        wn.az.resize((size_t)W * rows);
        wn.stamp.resize((size_t)W);
      }
      // the decode pool: T threads for the whole sequence (round 6; rounds 3-5 started a window's threads anew, 63 thread
      // starts per window), every PNG inflated and unfiltered straight into its slot of the page-locked window
      std::string decode_error;
      struct Pool {
        std::mutex mu;
        std::condition_variable cv_job, cv_done;
        std::vector<std::thread> threads;
        size_t f0 = 0;
        Win *wn = nullptr;
        std::atomic<int> next{0};
        int generation = 0, busy = 0;
        bool stop = false;
        std::atomic<long long> cpu_ns{0};
      } pool;
      auto decode_one = [&](size_t f0, Win *wn, int i, rsxhost::PngScratch &scratch) {
        const auto t0 = clk::now();
        int w = 0, h = 0;
        uint8_t *dst = wn->img + (size_t)i * ibytes;
        rsxhost::read_png_gray8_into(dir + "/" + files[f0 + (size_t)i], dst, ibytes, &w, &h, scratch);
        if (h != rows || w != w0) die(files[f0 + (size_t)i] + ": image shape changed");
        int64_t st = 0;
        std::memcpy(&st, dst, 8);  // little-endian int64 at bytes 0-7 of the first row
        if (st <= 0) st = std::atoll(files[f0 + (size_t)i].c_str());
        wn->stamp[(size_t)i] = st;
        for (int a = 0; a < rows; a++) {
          uint16_t cnt;
          std::memcpy(&cnt, dst + (size_t)a * w + 8, 2);
          wn->az[(size_t)i * rows + a] = (float)((double)cnt * 2.0 * M_PI / 5600.0);
        }
        pool.cpu_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count();
      };
      auto worker = [&]() {
        rsxhost::PngScratch scratch;
        int seen = 0;
        for (;;) {
          size_t f0;
          Win *wn;
          {
            std::unique_lock<std::mutex> lk(pool.mu);
            pool.cv_job.wait(lk, [&] { return pool.stop || pool.generation != seen; });
            if (pool.stop) return;
            seen = pool.generation;
            f0 = pool.f0;
            wn = pool.wn;
          }
          for (;;) {
            const int i = pool.next.fetch_add(1);
            if (i >= wn->n) break;
            try {
              decode_one(f0, wn, i, scratch);
            } catch (const std::exception &e) {
              std::lock_guard<std::mutex> lk(pool.mu);
              if (decode_error.empty()) decode_error = e.what();
            }
          }
          {
            std::lock_guard<std::mutex> lk(pool.mu);
            if (--pool.busy == 0) pool.cv_done.notify_all();
          }
        }
      };
      for (int t = 0; t < T; t++) pool.threads.emplace_back(worker);
      struct PoolStop {  // (also when something below throws)
        Pool &p;
        ~PoolStop() {
          {
            std::lock_guard<std::mutex> lk(p.mu);
            p.stop = true;
          }
          p.cv_job.notify_all();
          for (std::thread &t : p.threads) t.join();
        }
      } pool_stop{pool};
      // hands window [f0, f0 + W) to the pool and returns; wait_window() blocks until it is decoded
      auto start_window = [&](size_t f0, Win *wn) {
        wn->n = (int)std::min((size_t)W, files.size() - f0);
        std::lock_guard<std::mutex> lk(pool.mu);
        pool.f0 = f0;
        pool.wn = wn;
        pool.next.store(0);
        pool.cpu_ns.store(0);
        pool.busy = T;
        pool.generation++;
        pool.cv_job.notify_all();
      };
      auto wait_window = [&](Win *wn) {
        std::unique_lock<std::mutex> lk(pool.mu);
        pool.cv_done.wait(lk, [&] { return pool.busy == 0; });
        wn->decode_cpu_s = 1e-9 * (double)pool.cpu_ns.load();
      };
#ifdef RSX_WITH_ROS
      const bool want_xy = true;
#else
      const bool want_xy = rec != nullptr;
#endif
      const int max_xy = want_xy ? op.max_keypoints : 0;
      std::vector<float> xy(want_xy ? (size_t)W * max_xy * 2 : 0);
      std::vector<rsx_odometry_scan> res((size_t)W);
      double decode_cpu = 0, decode_wait = 0, push_s = 0;
      const auto t_all = clk::now();
      start_window(0, &wins[0]);
      size_t fi = 0;
      for (size_t f0 = 0, wi = 0; f0 < files.size(); f0 += (size_t)W, wi++) {
        const auto tw = clk::now();
        Win &wn = wins[wi & 1];
        wait_window(&wn);
        decode_wait += std::chrono::duration<double>(clk::now() - tw).count();
        if (!decode_error.empty()) die(decode_error);
        if (f0 + (size_t)W < files.size()) start_window(f0 + (size_t)W, &wins[(wi + 1) & 1]);
        decode_cpu += wn.decode_cpu_s;
        const auto tp = clk::now();
        check(rsx_odometry_push(odo, wn.img, wn.n, (int64_t)ibytes, w0, wn.az.data(), 1, res.data(), want_xy ? xy.data() : nullptr, max_xy),
              "rsx_odometry_push");
        push_s += std::chrono::duration<double>(clk::now() - tp).count();
        for (int i = 0; i < wn.n; i++, fi++) {
          compose(res[(size_t)i].reg);
          const int nk = std::min(res[(size_t)i].n_keypoints, op.max_keypoints);
          emit(fi, wn.stamp[(size_t)i], nk, (size_t)res[(size_t)i].n_matches, want_xy ? &xy[(size_t)i * max_xy * 2] : nullptr);
        }
#ifdef RSX_WITH_ROS
        if (!ros::ok()) break;
#endif
      }
      const double all_s = std::chrono::duration<double>(clk::now() - t_all).count();
      if (timing)
        std::fprintf(stderr,
                     "timing: scans=%zu window=%d decode_threads=%d decode_cpu_s=%.4f decode_ms_per_scan_per_thread=%.3f decode_wait_s=%.4f "
                     "pipeline_s=%.4f pipeline_scans_per_s=%.1f total_s=%.4f total_scans_per_s=%.1f\n",
                     files.size(), W, T, decode_cpu, 1e3 * decode_cpu / (double)files.size(), decode_wait, push_s, (double)files.size() / push_s,
                     all_s, (double)files.size() / all_s);
      for (Win &wn : wins) rsx_host_free_pinned(wn.img);
      rsx_odometry_destroy(odo);
      if (out != stdout) std::fclose(out);
      if (rec) std::fclose(rec);
      return 0;
    }

    // ---------------- one scan per call (round-2 loop; also the `nn` stand-in matcher) ----------------
    rsx_cen2019 *cen = nullptr;
    rsx_orora *reg = nullptr;
    rsx_frontend *fe = nullptr;
    const bool use_orb = matcher != "nn";
    check(rsx_orora_create(device, &reg), "rsx_orora_create");
    rsx_cen2019_params cp;
    rsx_cen2019_default_params(&cp);
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
        if (n_match > cap) {  // keep an evenly spread subset, and say so
          std::fprintf(stderr, "odometry: frame %zu: %zu matches exceed the solver's %zu, registering an evenly spread subset\n", fi, n_match, cap);
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
        rsx_orora_params rp;
        check(rsx_orora_default_params(&rp), "rsx_orora_default_params");
        if (use_pmc) rp.flags |= RSX_ORORA_PMC;
        check(rsx_orora_register_batch(reg, dst.data(), src.data(), offsets, 1, &rp, &r), "rsx_orora_register_batch");
        compose(r);
      }
      emit(fi, cur.stamp_ns, n, n_match, cur.xy.data());
#ifdef RSX_WITH_ROS
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
