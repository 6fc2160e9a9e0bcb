// scancontext/Scancontext.h -- drop-in replacement for the reference header of the same path
// (pgo/SC-A-LOAM/include/scancontext/Scancontext.h).  The class keeps the reference's public
// surface -- SCManager, makeAndSaveScancontextAndKeys, detectLoopClosureID, saveScancontextAndKeys,
// detectLoopClosureIDBetweenSession, getConstRefRecentSCD, setSCdistThres, the public hyper
// parameters that laserPosegraphOptimization.cpp reads (NUM_EXCLUDE_RECENT, PGO.cpp:558) -- and
// forwards every call to librsx.so (include/rsx.h), i.e. to the HIP kernels on the MI355X.
//
// This header is host glue only: no descriptor, key or distance is computed here.  It is written
// from the reference's interface (method names, argument meaning, -1 = "no loop"), not from its
// implementation.
//
// Build variants:
//   * with Eigen + PCL on the include path (the ROS build): the reference signatures
//     (pcl::PointCloud<pcl::PointXYZI>&, Eigen::MatrixXd) are available, so alaserPGO compiles
//     unchanged with `-I<this dir>` ahead of the reference include dir and `-lrsx`;
//   * without them (this container): the POD overloads below carry the same semantics.
//
// Differences a maintainer should know (DESIGN.md "boundary"):
//   * SCManager is internally synchronised (the reference races between PGO.cpp:492 and :561);
//   * failures of the GPU library throw std::runtime_error (the reference has no error path);
//   * the public std::vector members polarcontexts_ etc. are replaced by accessors
//     (descriptor(i), ringkey(i), sectorkey(i)) that read the HBM-resident database.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rsx.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define RSX_HAVE_EIGEN 1
#endif
#if __has_include(<pcl/point_cloud.h>) && __has_include(<pcl/point_types.h>)
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#define RSX_HAVE_PCL 1
using SCPointType = pcl::PointXYZI;  // Scancontext.h:39
#endif
#endif

using KeyMat = std::vector<std::vector<float>>;  // Scancontext.h:40

inline void coreImportTest(void);  // Scancontext.h:47

class SCManager {
 public:
  SCManager() = default;
  ~SCManager() {
    if (vg_) rsx_voxelgrid_destroy(vg_);
    if (h_) rsx_sc_destroy(h_);
  }
  SCManager(const SCManager &) = delete;
  SCManager &operator=(const SCManager &) = delete;

  // ---- user-side API (Scancontext.h:70-79) ----
#ifdef RSX_HAVE_PCL
  void makeAndSaveScancontextAndKeys(pcl::PointCloud<SCPointType> &scan_down) {
    // pcl::PointXYZI is 32 bytes: x,y,z at offsets 0,4,8
    makeAndSaveScancontextAndKeys(scan_down.points.empty() ? nullptr : &scan_down.points[0].x, scan_down.points.size(),
                                  sizeof(SCPointType));
  }
#endif
  // POD form: n points, stride_bytes apart, float x,y,z first
  void makeAndSaveScancontextAndKeys(const float *xyz, std::size_t n, std::size_t stride_bytes) {
    check(rsx_sc_add_points(handle(), xyz, n, stride_bytes, nullptr), "makeAndSaveScancontextAndKeys");
  }

  // Opt-in fusion of the caller's `downSizeFilterScancontext.filter(*thisKeyFrameDS)` with the build
  // (laserPosegraphOptimization.cpp:482-492): the raw keyframe goes in, the VoxelGrid downsample
  // (leaf as set at PGO.cpp:687-688) and the descriptor build both run on the GPU.
  void makeAndSaveScancontextAndKeysDownsampled(const float *xyz, std::size_t n, std::size_t stride_bytes, float leaf = 0.4f) {
    rsx_sc *h = handle();
    if (!vg_) check(rsx_voxelgrid_create(device_, &vg_), "rsx_voxelgrid_create");
    check(rsx_sc_add_points_downsampled(h, vg_, xyz, n, stride_bytes, leaf, nullptr), "makeAndSaveScancontextAndKeysDownsampled");
  }
#ifdef RSX_HAVE_PCL
  void makeAndSaveScancontextAndKeysDownsampled(pcl::PointCloud<SCPointType> &scan, float leaf = 0.4f) {
    makeAndSaveScancontextAndKeysDownsampled(scan.points.empty() ? nullptr : &scan.points[0].x, scan.points.size(),
                                             sizeof(SCPointType), leaf);
  }
#endif

  // int: nearest node index or -1, float: relative yaw [rad]
  std::pair<int, float> detectLoopClosureID(void) {
    int32_t id = -1, nn = 0;
    float yaw = 0.f;
    double md = 0;
    check(rsx_sc_detect_loop_closure(handle(), mode_, &id, &yaw, &md, &nn), "detectLoopClosureID");
    last_min_dist_ = md;
    last_nn_idx_ = nn;
    return {id, yaw};
  }

#ifdef RSX_HAVE_EIGEN
  void saveScancontextAndKeys(Eigen::MatrixXd scd) {  // Scancontext.h:76 (by value, like the reference)
    requireShape(scd.rows(), scd.cols());
    saveScancontextAndKeys(scd.data());
  }
  std::pair<int, float> detectLoopClosureIDBetweenSession(std::vector<float> &curr_key, Eigen::MatrixXd &curr_desc) {
    requireShape(curr_desc.rows(), curr_desc.cols());
    return detectLoopClosureIDBetweenSession(curr_key, curr_desc.data());
  }
  const Eigen::MatrixXd &getConstRefRecentSCD(void) {
    recent_ = Eigen::MatrixXd(RSX_SC_NUM_RING, RSX_SC_NUM_SECTOR);
    int64_t n = size();
    check(rsx_sc_get_descriptor(handle(), n - 1, recent_.data()), "getConstRefRecentSCD");
    return recent_;
  }
#endif
  // POD forms: 20x60 column-major doubles (Eigen::MatrixXd memory order)
  void saveScancontextAndKeys(const double *scd_colmajor) {
    check(rsx_sc_add_descriptor(handle(), scd_colmajor, nullptr), "saveScancontextAndKeys");
  }
  std::pair<int, float> detectLoopClosureIDBetweenSession(std::vector<float> &curr_key, const double *curr_desc_colmajor) {
    if (curr_key.size() != RSX_SC_NUM_RING) throw std::runtime_error("ring key must have 20 entries");
    int32_t id = -1, nn = 0;
    float yaw = 0.f;
    double md = 0;
    check(rsx_sc_detect_between_session(handle(), curr_key.data(), curr_desc_colmajor, &id, &yaw, &md, &nn),
          "detectLoopClosureIDBetweenSession");
    last_min_dist_ = md;
    last_nn_idx_ = nn;
    return {id, yaw};
  }

  // ---- hyper parameters (Scancontext.h:83-104): same names, same defaults ----
  const double LIDAR_HEIGHT = 2.0;
  const int PC_NUM_RING = RSX_SC_NUM_RING;
  const int PC_NUM_SECTOR = RSX_SC_NUM_SECTOR;
  const double PC_MAX_RADIUS = 80.0;
  const double PC_UNIT_SECTORANGLE = 360.0 / double(RSX_SC_NUM_SECTOR);
  const double PC_UNIT_RINGGAP = 80.0 / double(RSX_SC_NUM_RING);
  const int NUM_EXCLUDE_RECENT = 30;
  const int NUM_CANDIDATES_FROM_TREE = 3;
  const double SEARCH_RATIO = 0.1;
  double SC_DIST_THRES = 0.2;
  const int TREE_MAKING_PERIOD_ = 30;

  void setSCdistThres(double new_thres) {  // Scancontext.h:107
    SC_DIST_THRES = new_thres;
    if (h_) check(rsx_sc_set_dist_thres(h_, new_thres), "setSCdistThres");
  }

  // ---- extensions (opt-in; defaults reproduce the reference) ----
  // score the whole searchable prefix instead of the 3 ring-key neighbours
  void setExhaustive(bool on) { mode_ = on ? RSX_SC_MODE_EXHAUSTIVE : RSX_SC_MODE_CANDIDATE; }
  void setDevice(int device) { device_ = device; }  // before first use
  int64_t size() {
    int64_t n = 0;
    check(rsx_sc_size(handle(), &n), "size");
    return n;
  }
  double lastMinDist() const { return last_min_dist_; }  // the value of the reference's log line (SC.cpp:406,412)
  int lastNearestIndex() const { return last_nn_idx_; }
  std::vector<double> descriptor(int64_t i) {  // polarcontexts_[i], column-major 20x60
    std::vector<double> d(RSX_SC_DESC_SIZE);
    check(rsx_sc_get_descriptor(handle(), i, d.data()), "descriptor");
    return d;
  }
  std::vector<float> ringkey(int64_t i) {  // polarcontext_invkeys_mat_[i]
    std::vector<float> k(RSX_SC_NUM_RING);
    check(rsx_sc_get_ringkey(handle(), i, k.data()), "ringkey");
    return k;
  }
  std::vector<double> sectorkey(int64_t i) {  // polarcontext_vkeys_[i]
    std::vector<double> k(RSX_SC_NUM_SECTOR);
    check(rsx_sc_get_sectorkey(handle(), i, k.data()), "sectorkey");
    return k;
  }
  rsx_sc *handle() {
    if (!h_) {
      rsx_sc_params p;
      rsx_sc_default_params(&p);
      p.dist_thres = SC_DIST_THRES;
      p.device = device_;
      check(rsx_sc_create(&p, &h_), "SCManager (rsx_sc_create)");
    }
    return h_;
  }

 private:
  static void check(int status, const char *what) {
    if (status != RSX_OK)
      throw std::runtime_error(std::string(what) + ": rsx status " + std::to_string(status) + ": " + rsx_last_error_string());
  }
  static void requireShape(long rows, long cols) {
    if (rows != RSX_SC_NUM_RING || cols != RSX_SC_NUM_SECTOR) throw std::runtime_error("descriptor must be 20 x 60");
  }
  rsx_sc *h_ = nullptr;
  rsx_voxelgrid *vg_ = nullptr;
  int mode_ = RSX_SC_MODE_CANDIDATE;
  int device_ = 0;
  double last_min_dist_ = 0.0;
  int last_nn_idx_ = 0;
#ifdef RSX_HAVE_EIGEN
  Eigen::MatrixXd recent_;
#endif
};

#include <iostream>
inline void coreImportTest(void) { std::cout << "scancontext lib (rsx / MI355X) is successfully imported." << std::endl; }
