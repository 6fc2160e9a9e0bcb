// sc_shim_demo.cpp -- exercises the SCManager shim exactly the way laserPosegraphOptimization.cpp
// does (PGO.cpp:99,492,558-571,685): one writer adding keyframes, one reader polling for loops.
// Build: g++ -std=c++17 -I../../include -Iscancontext/.. sc_shim_demo.cpp -L.. -lrsx -pthread
// Prints one line per detected loop in the reference's own format (PGO.cpp:566).
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <iostream>
#include <random>
#include <thread>
#include <vector>

#include "scancontext/Scancontext.h"

SCManager scManager;  // PGO.cpp:99

struct Pt {
  float x, y, z, intensity, pad[4];  // pcl::PointXYZI layout (32 B)
};

static std::vector<Pt> synth_cloud(std::mt19937 &rng, float yaw, const std::vector<Pt> *base) {
  std::uniform_real_distribution<float> ur(2.f, 80.f), ua(0.f, 6.2831853f);
  std::normal_distribution<float> jit(0.f, 0.05f);
  std::vector<Pt> c;
  if (base) {
    const float cs = std::cos(yaw), sn = std::sin(yaw);
    for (const Pt &p : *base) c.push_back(Pt{cs * p.x - sn * p.y + jit(rng), sn * p.x + cs * p.y + jit(rng), 0.f, 0.f, {}});
  } else {
    for (int i = 0; i < 1500; i++) {
      float r = ur(rng), a = ua(rng);
      c.push_back(Pt{r * std::cos(a), r * std::sin(a), 0.f, 0.f, {}});
    }
  }
  return c;
}

// --replay <file> [--devices a,b,..] [--exhaustive]: feed recorded clouds (int32 n_clouds, then per cloud
// int32 n_points + n_points x {x,y,z,intensity} floats) through the shim exactly like process_pg / process_lcd do,
// one detectLoopClosureID() per keyframe.  stdout = the shim's own "[Loop found] / [Not loop]" lines
// (Scancontext.cpp:406,412) followed by one "RESULT i loop_id yaw" line per keyframe; tests/test_gpu_host.py
// compares both with the reference build's output on the same clouds.
static int replay(const char *path, const std::vector<int> &devices, bool exhaustive) {
  FILE *f = std::fopen(path, "rb");
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", path);
    return 1;
  }
  int32_t n_clouds = 0;
  if (std::fread(&n_clouds, 4, 1, f) != 1) return 1;
  SCManager sc;
  sc.setSCdistThres(0.45);
  if (devices.size() > 1) sc.setDevices(devices);
  else if (devices.size() == 1) sc.setDevice(devices[0]);
  sc.setExhaustive(exhaustive);
  std::vector<float> raw;
  std::vector<Pt> cloud;
  for (int i = 0; i < n_clouds; i++) {
    int32_t n = 0;
    if (std::fread(&n, 4, 1, f) != 1) return 1;
    raw.resize((size_t)n * 4);
    if (n && std::fread(raw.data(), 16, (size_t)n, f) != (size_t)n) return 1;
    cloud.clear();
    for (int j = 0; j < n; j++) cloud.push_back(Pt{raw[4 * j], raw[4 * j + 1], raw[4 * j + 2], raw[4 * j + 3], {}});
    sc.makeAndSaveScancontextAndKeys(n ? &cloud[0].x : nullptr, cloud.size(), sizeof(Pt));
    auto r = sc.detectLoopClosureID();
    std::printf("RESULT %d %d %.9g\n", i, r.first, (double)r.second);
    std::fflush(stdout);
  }
  std::fclose(f);
  // the public helpers (Scancontext.h:60-66) on the last two keyframes
  std::vector<double> a = sc.descriptor(n_clouds - 1), b = sc.descriptor(n_clouds - 2);
  double rk[20], vk1[60], vk2[60];
  sc.makeRingkeyFromScancontext(a.data(), rk);
  sc.makeSectorkeyFromScancontext(a.data(), vk1);
  sc.makeSectorkeyFromScancontext(b.data(), vk2);
  auto d = sc.distanceBtnScanContext(a.data(), b.data());
  std::printf("HELPERS %.17g %.17g %d %.17g %d %.17g\n", rk[3], vk1[7], sc.fastAlignUsingVkey(vk1, vk2), d.first, d.second,
              sc.distDirectSC(a.data(), b.data()));
  // the reference's public data members as views of the GPU database (Scancontext.h:110-115)
  {
    const auto &last = sc.polarcontexts_.back();
    double sum = 0.0;
    for (int e = 0; e < RSX_SC_DESC_SIZE; e++) sum += last.data()[e];
    std::printf("MEMBERS %zu %.17g %.9g %.17g %.17g\n", sc.polarcontexts_.size(), sum, (double)sc.polarcontext_invkeys_mat_[(size_t)n_clouds - 1][3],
                sc.polarcontext_invkeys_[(size_t)n_clouds - 1].data()[3], sc.polarcontext_vkeys_.at((size_t)n_clouds - 1).data()[7]);
  }
  return 0;
}

// --stream <file> [--every N]: the streaming-SLAM emulation of bench.py (BASELINE configs[3]) from C++ -- every keyframe of the
// file goes into TWO managers (the reference's candidate detector and the exhaustive one), every N-th keyframe both detect.
// The clouds are read into memory first; timed: the loop of makeAndSaveScancontextAndKeys / detectLoopClosureID calls only.
// stdout: one "DET i cand_id cand_yaw exh_id exh_yaw" line per detection, then "STREAM keyframes=.. detections=.. seconds=.. keyframes_per_sec=..".
static int stream(const char *path, int every) {
  FILE *f = std::fopen(path, "rb");
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", path);
    return 1;
  }
  int32_t n_clouds = 0;
  if (std::fread(&n_clouds, 4, 1, f) != 1) return 1;
  std::vector<std::vector<Pt>> clouds((size_t)n_clouds);
  std::vector<float> raw;
  for (int i = 0; i < n_clouds; i++) {
    int32_t n = 0;
    if (std::fread(&n, 4, 1, f) != 1) return 1;
    raw.resize((size_t)n * 4);
    if (n && std::fread(raw.data(), 16, (size_t)n, f) != (size_t)n) return 1;
    for (int j = 0; j < n; j++) clouds[(size_t)i].push_back(Pt{raw[4 * j], raw[4 * j + 1], raw[4 * j + 2], raw[4 * j + 3], {}});
  }
  std::fclose(f);
  SCManager cand, exh;
  cand.setSCdistThres(0.45);
  exh.setSCdistThres(0.45);
  exh.setExhaustive(true);
  cand.setVerbose(false);  // (the reference's "[Loop found] / [Not loop]" line per detection: stdout is not what is timed here)
  exh.setVerbose(false);
  cand.handle();
  exh.handle();
  struct Det {
    int i, c_id, e_id;
    float c_yaw, e_yaw;
  };
  std::vector<Det> dets;
  dets.reserve((size_t)n_clouds / (every > 0 ? every : 1) + 1);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n_clouds; i++) {
    const std::vector<Pt> &c = clouds[(size_t)i];
    cand.makeAndSaveScancontextAndKeys(c.empty() ? nullptr : &c[0].x, c.size(), sizeof(Pt));
    exh.makeAndSaveScancontextAndKeys(c.empty() ? nullptr : &c[0].x, c.size(), sizeof(Pt));
    if (every > 0 && i % every == every - 1) {
      const auto a = cand.detectLoopClosureID(), b = exh.detectLoopClosureID();
      dets.push_back(Det{i, a.first, b.first, a.second, b.second});
    }
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (const Det &d : dets) std::printf("DET %d %d %.9g %d %.9g\n", d.i, d.c_id, (double)d.c_yaw, d.e_id, (double)d.e_yaw);
  std::printf("STREAM keyframes=%d detections=%zu seconds=%.6f keyframes_per_sec=%.1f\n", n_clouds, dets.size(), dt, n_clouds / dt);
  return 0;
}

int main(int argc, char **argv) {
  try {
    if (argc >= 3 && std::string(argv[1]) == "--stream") {
      int every = 4;
      for (int i = 3; i + 1 < argc; i++)
        if (std::string(argv[i]) == "--every") every = std::atoi(argv[i + 1]);
      return stream(argv[2], every);
    }
    if (argc >= 3 && std::string(argv[1]) == "--replay") {
      std::vector<int> devices;
      bool exhaustive = false;
      for (int i = 3; i < argc; i++) {
        if (std::string(argv[i]) == "--exhaustive") exhaustive = true;
        if (std::string(argv[i]) == "--devices" && i + 1 < argc) {
          for (const char *p = argv[++i]; *p;) {
            devices.push_back(std::atoi(p));
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
          }
        }
      }
      return replay(argv[2], devices, exhaustive);
    }
    coreImportTest();
    scManager.setSCdistThres(0.45);  // PGO.cpp:677,685 + sc_pgo.launch:4
    scManager.handle();              // create the GPU handle up front: no device -> fail here, loudly
    std::mt19937 rng(1234);
    std::vector<std::vector<Pt>> clouds;
    std::atomic<bool> done{false};
    std::atomic<int> loops{0};
    std::thread lc_detection([&] {  // process_lcd, PGO.cpp:573-585 (polling faster than 1 Hz for the demo)
      while (!done) {
        if (scManager.size() >= scManager.NUM_EXCLUDE_RECENT) {  // PGO.cpp:558
          auto r = scManager.detectLoopClosureID();              // PGO.cpp:561
          if (r.first != -1) {
            std::cout << "Loop detected! - between " << r.first << " and " << scManager.size() - 1 << "" << std::endl;
            loops++;
          }
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
    });
    for (int i = 0; i < 120; i++) {  // process_pg, PGO.cpp:492
      const bool revisit = i >= 60 && i % 10 == 0;
      clouds.push_back(revisit ? synth_cloud(rng, 0.5236f, &clouds[i - 55]) : synth_cloud(rng, 0.f, nullptr));
      // odd keyframes take the fused VoxelGrid(0.4 m) + build path (PGO.cpp:482-492 in one call)
      if (i & 1) scManager.makeAndSaveScancontextAndKeysDownsampled(&clouds.back()[0].x, clouds.back().size(), sizeof(Pt));
      else scManager.makeAndSaveScancontextAndKeys(&clouds.back()[0].x, clouds.back().size(), sizeof(Pt));
      std::this_thread::sleep_for(std::chrono::milliseconds(3));
    }
    done = true;
    lc_detection.join();
    std::printf("keyframes=%lld loops_reported=%d\n", (long long)scManager.size(), loops.load());
    return loops > 0 ? 0 : 2;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "fatal: %s\n", e.what());  // e.g. no GPU: the shim has no CPU fallback
    return 1;
  }
}
