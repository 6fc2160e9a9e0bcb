// sc_shim_demo.cpp -- exercises the SCManager shim exactly the way laserPosegraphOptimization.cpp
// does (PGO.cpp:99,492,558-571,685): one writer adding keyframes, one reader polling for loops.
// Build: g++ -std=c++17 -I../../include -Iscancontext/.. sc_shim_demo.cpp -L.. -lrsx -pthread
// Prints one line per detected loop in the reference's own format (PGO.cpp:566).
#include <atomic>
#include <cmath>
#include <cstdio>
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

int main() {
  try {
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
