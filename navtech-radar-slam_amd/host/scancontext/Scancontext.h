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
//   * SCManager is internally synchronised (the reference races between PGO.cpp:492 and :561): the GPU
//     handle is created once under a lock, every call locks the handle inside librsx, and the shim's own
//     cached values (last log values, recent SCD) are per thread or guarded;
//   * failures of the GPU library throw std::runtime_error (the reference has no error path);
//   * the public std::vector members polarcontexts_, polarcontext_invkeys_, polarcontext_vkeys_ and
//     polarcontext_invkeys_mat_ (Scancontext.h:110-115) exist as READ-ONLY views of the HBM-resident database with the
//     std::vector reading interface (size, [], at, front, back, iteration): elements are fetched on first access and
//     cached; code that writes to them does not compile.  descriptor(i) / ringkey(i) / sectorkey(i) return copies;
//   * the database stores fp32: saveScancontextAndKeys(MatrixXd) rounds a descriptor that is not
//     fp32-exact (every descriptor makeScancontext builds is) and remembers the largest rounding error
//     (lastImportRounding()); setStrictImport(true) makes such a descriptor an error instead;
//   * setDevice / setDevices / setExhaustive must be called before the first use (they throw afterwards);
//   * detectLoopClosureID prints the reference's "[Loop found] / [Not loop] Nearest distance ..." line
//     (Scancontext.cpp:406,412), including its std::cout.precision(3) side effect; setVerbose(false) mutes it.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <iostream>
#include <map>
#include <mutex>
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

// Read-only view with the reading interface of the std::vector the reference keeps in host memory
// (Scancontext.h:110-115), over the HBM-resident database: an element is fetched on first access and cached (the
// database is append-only, so an element never changes); references stay valid for the life of the SCManager.
template <typename T>
class RsxMirrorVector {
 public:
  using value_type = T;
  using size_type = std::size_t;
  RsxMirrorVector(std::function<int64_t()> size, std::function<T(int64_t)> fetch) : size_(std::move(size)), fetch_(std::move(fetch)) {}
  RsxMirrorVector(const RsxMirrorVector &) = delete;
  RsxMirrorVector &operator=(const RsxMirrorVector &) = delete;
  size_type size() const { return (size_type)size_(); }
  bool empty() const { return size() == 0; }
  const T &operator[](size_type i) const {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = cache_.find(i);
    if (it == cache_.end()) it = cache_.emplace(i, fetch_((int64_t)i)).first;
    return it->second;  // std::map nodes do not move
  }
  const T &at(size_type i) const {
    if (i >= size()) throw std::out_of_range("SCManager database view: index out of range");
    return (*this)[i];
  }
  const T &front() const { return (*this)[0]; }
  const T &back() const { return (*this)[size() - 1]; }
  class const_iterator {
   public:
    const_iterator(const RsxMirrorVector *v, size_type i) : v_(v), i_(i) {}
    const T &operator*() const { return (*v_)[i_]; }
    const T *operator->() const { return &(*v_)[i_]; }
    const_iterator &operator++() { ++i_; return *this; }
    bool operator!=(const const_iterator &o) const { return i_ != o.i_; }
    bool operator==(const const_iterator &o) const { return i_ == o.i_; }
   private:
    const RsxMirrorVector *v_;
    size_type i_;
  };
  const_iterator begin() const { return const_iterator(this, 0); }
  const_iterator end() const { return const_iterator(this, size()); }

 private:
  std::function<int64_t()> size_;
  std::function<T(int64_t)> fetch_;
  mutable std::mutex mu_;
  mutable std::map<size_type, T> cache_;
};

class SCManager {
 public:
  SCManager() = default;
  ~SCManager() {
    if (vg_) rsx_voxelgrid_destroy(vg_);
    if (h_) rsx_sc_destroy(h_);
    if (hs_) rsx_scs_destroy(hs_);
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
    if (sharded()) check(rsx_scs_add_points(shardedHandle(), xyz, n, stride_bytes, nullptr), "makeAndSaveScancontextAndKeys");
    else check(rsx_sc_add_points(handle(), xyz, n, stride_bytes, nullptr), "makeAndSaveScancontextAndKeys");
  }

  // Opt-in fusion of the caller's `downSizeFilterScancontext.filter(*thisKeyFrameDS)` with the build
  // (laserPosegraphOptimization.cpp:482-492): the raw keyframe goes in, the VoxelGrid downsample
  // (leaf as set at PGO.cpp:687-688) and the descriptor build both run on the GPU.
  // Opt-in: the keyframe-cloud store of the pose-graph node (keyframeLaserClouds, PGO.cpp:487) on the GPU.  Once attached,
  // makeAndSaveScancontextAndKeysDownsampled also appends the downsampled cloud to it -- what PGO.cpp:482-492 does in three
  // statements -- so that rsx_loop_verify / rsx_kfstore_build_map (doICPVirtualRelative, pubMap) find every keyframe in
  // HBM.  The store stays the caller's; it must live on the (first) device of this manager.
  void attachKeyframeStore(rsx_kfstore *kf) {
    std::lock_guard<std::mutex> lk(mu_);
    kf_ = kf;
  }
  void makeAndSaveScancontextAndKeysDownsampled(const float *xyz, std::size_t n, std::size_t stride_bytes, float leaf = 0.4f) {
    const int32_t ioff = stride_bytes >= 20 ? 16 : -1;  // pcl::PointXYZI keeps its intensity at byte 16
    if (sharded()) {
      // several devices: the downsample runs on the first one, the centroids (x, y, z, intensity float4) come back once
      // and every shard gets the same cloud -- the descriptor is the one a single device would build
      rsx_scs *hs = shardedHandle();
      std::lock_guard<std::mutex> lk(mu_);
      if (!vg_) check(rsx_voxelgrid_create(devices_[0], &vg_), "rsx_voxelgrid_create");
      ds_.resize(4 * (n ? n : 1));
      int64_t m = 0;
      check(rsx_voxelgrid_filter(vg_, xyz, n, stride_bytes, ioff, leaf, ds_.data(), (int64_t)(n ? n : 1), &m), "rsx_voxelgrid_filter");
      if (kf_) check(rsx_kfstore_add(kf_, ds_.data(), (std::size_t)m, 16, 12, nullptr), "rsx_kfstore_add");
      check(rsx_scs_add_points(hs, ds_.data(), (std::size_t)m, 16, nullptr), "makeAndSaveScancontextAndKeysDownsampled");
      return;
    }
    rsx_sc *h = handle();
    rsx_kfstore *kf = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (!vg_) check(rsx_voxelgrid_create(device_, &vg_), "rsx_voxelgrid_create");
      kf = kf_;
    }
    if (kf) check(rsx_sc_add_keyframe(h, vg_, kf, xyz, n, stride_bytes, ioff, leaf, nullptr), "makeAndSaveScancontextAndKeysDownsampled");
    else check(rsx_sc_add_points_downsampled(h, vg_, xyz, n, stride_bytes, leaf, nullptr), "makeAndSaveScancontextAndKeysDownsampled");
  }
#ifdef RSX_HAVE_PCL
  void makeAndSaveScancontextAndKeysDownsampled(pcl::PointCloud<SCPointType> &scan, float leaf = 0.4f) {
    makeAndSaveScancontextAndKeysDownsampled(scan.points.empty() ? nullptr : &scan.points[0].x, scan.points.size(),
                                             sizeof(SCPointType), leaf);
  }
#endif

  // int: nearest node index or -1, float: relative yaw [rad]
  std::pair<int, float> detectLoopClosureID(void) {
    rsx_sc_detection d;
    if (sharded()) check(rsx_scs_detect_loop_closure(shardedHandle(), &d), "detectLoopClosureID");
    else check(rsx_sc_detect_loop_closure_ex(handle(), mode_, &d), "detectLoopClosureID");
    {
      std::lock_guard<std::mutex> lk(mu_);
      last_min_dist_ = d.min_dist;
      last_nn_idx_ = d.nn_idx;
    }
    if (verbose_ && d.searched) {  // Scancontext.cpp:401-414, character for character
      using std::cout;
      using std::endl;
      if (d.min_dist < d.dist_thres) {
        cout << "[Loop found] Nearest distance: " << d.min_dist << " btn " << d.query_idx << " and " << d.nn_idx << "." << endl;
      } else {
        std::cout.precision(3);
        cout << "[Not loop] Nearest distance: " << d.min_dist << " btn " << d.query_idx << " and " << d.nn_idx << "." << endl;
      }
    }
    return {d.loop_id, d.yaw_diff_rad};
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
  // a reference into per-thread storage: valid until the calling thread's next call (the reference returns a
  // reference into its vector, Scancontext.cpp:230-233, which the next push_back may invalidate just the same)
  const Eigen::MatrixXd &getConstRefRecentSCD(void) {
    static thread_local Eigen::MatrixXd recent;
    recent = Eigen::MatrixXd(RSX_SC_NUM_RING, RSX_SC_NUM_SECTOR);
    getRecentSCD(recent.data());
    return recent;
  }
  // ---- public helpers of the reference (Scancontext.h:60-66), computed on the GPU ----
#ifdef RSX_HAVE_PCL
  Eigen::MatrixXd makeScancontext(pcl::PointCloud<SCPointType> &scan_down) {
    Eigen::MatrixXd d(RSX_SC_NUM_RING, RSX_SC_NUM_SECTOR);
    makeScancontext(scan_down.points.empty() ? nullptr : &scan_down.points[0].x, scan_down.points.size(), sizeof(SCPointType), d.data());
    return d;
  }
#endif
  Eigen::MatrixXd makeRingkeyFromScancontext(Eigen::MatrixXd &desc) {
    requireShape(desc.rows(), desc.cols());
    Eigen::MatrixXd k(RSX_SC_NUM_RING, 1);
    makeRingkeyFromScancontext(desc.data(), k.data());
    return k;
  }
  Eigen::MatrixXd makeSectorkeyFromScancontext(Eigen::MatrixXd &desc) {
    requireShape(desc.rows(), desc.cols());
    Eigen::MatrixXd k(1, RSX_SC_NUM_SECTOR);
    makeSectorkeyFromScancontext(desc.data(), k.data());
    return k;
  }
  int fastAlignUsingVkey(Eigen::MatrixXd &vkey1, Eigen::MatrixXd &vkey2) {
    if (vkey1.size() != RSX_SC_NUM_SECTOR || vkey2.size() != RSX_SC_NUM_SECTOR) throw std::runtime_error("sector keys must have 60 entries");
    return fastAlignUsingVkey(vkey1.data(), vkey2.data());
  }
  double distDirectSC(Eigen::MatrixXd &sc1, Eigen::MatrixXd &sc2) {
    requireShape(sc1.rows(), sc1.cols());
    requireShape(sc2.rows(), sc2.cols());
    return distDirectSC(sc1.data(), sc2.data());
  }
  std::pair<double, int> distanceBtnScanContext(Eigen::MatrixXd &sc1, Eigen::MatrixXd &sc2) {
    requireShape(sc1.rows(), sc1.cols());
    requireShape(sc2.rows(), sc2.cols());
    return distanceBtnScanContext(sc1.data(), sc2.data());
  }
#endif
  // POD forms of the helpers: descriptors are 20 x 60 column-major doubles, keys 20 / 60 doubles
  void makeScancontext(const float *xyz, std::size_t n, std::size_t stride_bytes, double *out_desc) {
    check(rsx_sc_make_scancontext(anyHandle(), xyz, n, stride_bytes, out_desc), "makeScancontext");
  }
  void makeRingkeyFromScancontext(const double *desc, double *out20) {
    check(rsx_sc_make_keys(anyHandle(), desc, out20, nullptr), "makeRingkeyFromScancontext");
  }
  void makeSectorkeyFromScancontext(const double *desc, double *out60) {
    check(rsx_sc_make_keys(anyHandle(), desc, nullptr, out60), "makeSectorkeyFromScancontext");
  }
  int fastAlignUsingVkey(const double *vkey1, const double *vkey2) {
    int32_t k = 0;
    check(rsx_sc_fast_align(anyHandle(), vkey1, vkey2, &k), "fastAlignUsingVkey");
    return k;
  }
  double distDirectSC(const double *sc1, const double *sc2) {
    double d = 0;
    check(rsx_sc_dist_direct(anyHandle(), sc1, sc2, &d), "distDirectSC");
    return d;
  }
  std::pair<double, int> distanceBtnScanContext(const double *sc1, const double *sc2) {
    double d = 0;
    int32_t k = 0;
    check(rsx_sc_distance(anyHandle(), sc1, sc2, &d, &k), "distanceBtnScanContext");
    return {d, k};
  }
  void getRecentSCD(double *out_colmajor) {  // polarcontexts_.back()
    const int64_t n = size();
    if (sharded()) check(rsx_scs_get_descriptor(shardedHandle(), n - 1, out_colmajor), "getConstRefRecentSCD");
    else check(rsx_sc_get_descriptor(handle(), n - 1, out_colmajor), "getConstRefRecentSCD");
  }
  // POD forms: 20x60 column-major doubles (Eigen::MatrixXd memory order)
  void saveScancontextAndKeys(const double *scd_colmajor) {
    if (sharded()) throw std::runtime_error("saveScancontextAndKeys: single-device handles only");
    if (strict_import_) {
      check(rsx_sc_add_descriptor(handle(), scd_colmajor, nullptr), "saveScancontextAndKeys");
      return;
    }
    double err = 0.0;
    check(rsx_sc_add_descriptor_rounded(handle(), scd_colmajor, nullptr, &err), "saveScancontextAndKeys");
    std::lock_guard<std::mutex> lk(mu_);
    if (err > last_import_rounding_) last_import_rounding_ = err;
  }
  std::pair<int, float> detectLoopClosureIDBetweenSession(std::vector<float> &curr_key, const double *curr_desc_colmajor) {
    if (curr_key.size() != RSX_SC_NUM_RING) throw std::runtime_error("ring key must have 20 entries");
    int32_t id = -1, nn = 0;
    float yaw = 0.f;
    double md = 0;
    check(rsx_sc_detect_between_session(handle(), curr_key.data(), curr_desc_colmajor, &id, &yaw, &md, &nn),
          "detectLoopClosureIDBetweenSession");
    std::lock_guard<std::mutex> lk(mu_);
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

  // ---- the reference's public data members (Scancontext.h:110-115) as read-only views of the GPU database ----
#ifdef RSX_HAVE_EIGEN
  using SCMat = Eigen::MatrixXd;  // polarcontexts_[i] is 20 x 60, polarcontext_invkeys_[i] 20 x 1, polarcontext_vkeys_[i] 1 x 60
#else
  using SCMat = std::vector<double>;  // the same numbers, column-major
#endif
  RsxMirrorVector<SCMat> polarcontexts_{[this] { return size(); }, [this](int64_t i) { return mirrorMat(i, 0); }};
  RsxMirrorVector<SCMat> polarcontext_invkeys_{[this] { return size(); }, [this](int64_t i) { return mirrorMat(i, 1); }};
  RsxMirrorVector<SCMat> polarcontext_vkeys_{[this] { return size(); }, [this](int64_t i) { return mirrorMat(i, 2); }};
  RsxMirrorVector<std::vector<float>> polarcontext_invkeys_mat_{[this] { return size(); }, [this](int64_t i) {
                                                                   const SCMat k = mirrorMat(i, 1);  // eig2stdvec: double -> float
                                                                   std::vector<float> f(RSX_SC_NUM_RING);
                                                                   for (int r = 0; r < RSX_SC_NUM_RING; r++) f[(std::size_t)r] = (float)k.data()[r];
                                                                   return f;
                                                                 }};

  void setSCdistThres(double new_thres) {  // Scancontext.h:107
    std::lock_guard<std::mutex> lk(mu_);
    SC_DIST_THRES = new_thres;
    if (h_) check(rsx_sc_set_dist_thres(h_, new_thres), "setSCdistThres");
    if (hs_) check(rsx_scs_set_dist_thres(hs_, new_thres), "setSCdistThres");
  }

  // ---- extensions (opt-in; defaults reproduce the reference) ----
  // score the whole searchable prefix instead of the 3 ring-key neighbours
  void setExhaustive(bool on) { mode_ = on ? RSX_SC_MODE_EXHAUSTIVE : RSX_SC_MODE_CANDIDATE; }
  void setVerbose(bool on) { verbose_ = on; }
  void setStrictImport(bool on) { strict_import_ = on; }
  // how the reference this shim replaces was BUILT: the order of the sums it takes through Eigen (rsx.h RSX_SC_SUM_*).  The
  // default is the reference's own CMakeLists.txt (x86-64, SSE2); a catkin workspace compiled with -march=native wants
  // RSX_SC_SUM_EIGEN_AVX_FMA for bit-identical loop decisions.  Before first use.
  void setSumOrder(int order) {
    std::lock_guard<std::mutex> lk(mu_);
    if (h_ || hs_) throw std::runtime_error("setSumOrder after the GPU handle was created");
    if (order < RSX_SC_SUM_EIGEN_SSE2 || order > RSX_SC_SUM_EIGEN34_AVX_FMA) throw std::runtime_error("setSumOrder: unknown order");
    sum_order_ = order;
  }
  void setDevice(int device) {  // before first use
    std::lock_guard<std::mutex> lk(mu_);
    if (h_ || hs_) throw std::runtime_error("setDevice after the GPU handle was created");
    device_ = device;
  }
  // shard the database over several GPUs of this node (keyframe i on devices[i % n]); the detector then scores the
  // whole searchable prefix (exhaustive mode) on all of them.  Before first use.
  // query_groups: devices.size() = query_groups x DB shards (rsx_scs_create_layout): 1 = every device a shard of one
  // database copy (the detector's single query is then scored by all devices at once); more groups only pay for
  // batched queries (query()).  rccl_exchange: ncclAllGather instead of peer copies between the shards of a group.
  void setDevices(const std::vector<int> &devices, int query_groups = 1, bool rccl_exchange = false) {
    std::lock_guard<std::mutex> lk(mu_);
    if (h_ || hs_) throw std::runtime_error("setDevices after the GPU handle was created");
    if (devices.empty()) throw std::runtime_error("setDevices: empty list");
    if (query_groups < 1 || devices.size() % (std::size_t)query_groups) throw std::runtime_error("setDevices: query_groups must divide the device count");
    devices_.assign(devices.begin(), devices.end());
    device_ = devices[0];
    query_groups_ = query_groups;
    rccl_exchange_ = rccl_exchange;
  }
  bool sharded() const { return devices_.size() > 1; }
  int64_t size() {
    int64_t n = 0;
    if (sharded()) check(rsx_scs_size(shardedHandle(), &n), "size");
    else check(rsx_sc_size(handle(), &n), "size");
    return n;
  }
  double lastMinDist() {  // the value of the reference's log line (SC.cpp:406,412)
    std::lock_guard<std::mutex> lk(mu_);
    return last_min_dist_;
  }
  int lastNearestIndex() {
    std::lock_guard<std::mutex> lk(mu_);
    return last_nn_idx_;
  }
  double lastImportRounding() {
    std::lock_guard<std::mutex> lk(mu_);
    return last_import_rounding_;
  }
  std::vector<double> descriptor(int64_t i) {  // polarcontexts_[i], column-major 20x60
    std::vector<double> d(RSX_SC_DESC_SIZE);
    if (sharded()) check(rsx_scs_get_descriptor(shardedHandle(), i, d.data()), "descriptor");
    else check(rsx_sc_get_descriptor(handle(), i, d.data()), "descriptor");
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
  // batched exhaustive query (the north-star path), host buffers: nq f32 sector-major descriptors -> nq x k hits
  void query(const float *q_descs, int32_t nq, int32_t k, int64_t n_eligible, rsx_sc_hit *out) {
    if (sharded()) check(rsx_scs_query(shardedHandle(), q_descs, nq, k, n_eligible, out), "query");
    else check(rsx_sc_query(handle(), q_descs, nq, k, n_eligible, out), "query");
  }
  rsx_sc *handle() {
    std::lock_guard<std::mutex> lk(mu_);
    if (sharded()) throw std::runtime_error("single-device call on a multi-device SCManager");
    if (!h_) {
      rsx_sc_params p;
      rsx_sc_default_params(&p);
      p.dist_thres = SC_DIST_THRES;
      p.device = device_;
      p.sum_order = sum_order_;
      check(rsx_sc_create(&p, &h_), "SCManager (rsx_sc_create)");
    }
    return h_;
  }
  rsx_scs *shardedHandle() {
    std::lock_guard<std::mutex> lk(mu_);
    if (!hs_) {
      rsx_sc_params p;
      rsx_sc_default_params(&p);
      p.dist_thres = SC_DIST_THRES;
      p.sum_order = sum_order_;
      std::vector<int32_t> dev(devices_.begin(), devices_.end());
      check(rsx_scs_create_layout(&p, dev.data(), (int32_t)dev.size(), query_groups_, rccl_exchange_ ? RSX_SCS_EXCHANGE_RCCL : RSX_SCS_EXCHANGE_PEER_COPY,
                                  &hs_),
            "SCManager (rsx_scs_create_layout)");
    }
    return hs_;
  }

 private:
  static void check(int status, const char *what) {
    if (status != RSX_OK)
      throw std::runtime_error(std::string(what) + ": rsx status " + std::to_string(status) + ": " + rsx_last_error_string());
  }
  static void requireShape(long rows, long cols) {
    if (rows != RSX_SC_NUM_RING || cols != RSX_SC_NUM_SECTOR) throw std::runtime_error("descriptor must be 20 x 60");
  }
  // the stateless helpers need a device: the single-device handle, or in multi-device mode a small helper handle
  rsx_sc *anyHandle() {
    if (!sharded()) return handle();
    std::lock_guard<std::mutex> lk(mu_);
    if (!h_) {
      rsx_sc_params p;
      rsx_sc_default_params(&p);
      p.device = device_;
      p.sum_order = sum_order_;
      p.capacity_hint = 32;
      check(rsx_sc_create(&p, &h_), "SCManager (helper handle)");
    }
    return h_;
  }
  // element i of a mirrored member: 0 = the descriptor, 1 = its ring key, 2 = its sector key (keys through the GPU helpers)
  SCMat mirrorMat(int64_t i, int what) {
    const std::vector<double> d = descriptor(i);
    const int rows = what == 2 ? 1 : RSX_SC_NUM_RING, cols = what == 0 ? RSX_SC_NUM_SECTOR : (what == 1 ? 1 : RSX_SC_NUM_SECTOR);
#ifdef RSX_HAVE_EIGEN
    SCMat m(rows, cols);
#else
    SCMat m((std::size_t)rows * cols);
#endif
    if (what == 0) for (int e = 0; e < RSX_SC_DESC_SIZE; e++) m.data()[e] = d[(std::size_t)e];
    else if (what == 1) makeRingkeyFromScancontext(d.data(), m.data());
    else makeSectorkeyFromScancontext(d.data(), m.data());
    return m;
  }
  std::mutex mu_;  // guards handle creation and the cached values below
  rsx_sc *h_ = nullptr;
  rsx_scs *hs_ = nullptr;
  rsx_voxelgrid *vg_ = nullptr;
  rsx_kfstore *kf_ = nullptr;  // not owned (attachKeyframeStore)
  std::vector<float> ds_;  // downsampled cloud of the multi-device path
  int mode_ = RSX_SC_MODE_CANDIDATE;
  int device_ = 0;
  int sum_order_ = RSX_SC_SUM_EIGEN_SSE2;
  std::vector<int> devices_;
  int query_groups_ = 1;
  bool rccl_exchange_ = false;
  bool verbose_ = true, strict_import_ = false;
  double last_min_dist_ = 0.0, last_import_rounding_ = 0.0;
  int last_nn_idx_ = 0;
};

inline void coreImportTest(void) { std::cout << "scancontext lib (rsx / MI355X) is successfully imported." << std::endl; }
