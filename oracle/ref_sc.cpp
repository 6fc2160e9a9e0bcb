/*
 * oracle/ref_sc.cpp -- extern "C" wrapper around the REFERENCE'S OWN Scancontext.cpp.
 * TEST INFRASTRUCTURE ONLY.
 *
 * Nothing is copied: the reference source is #included from where it lies under /root/reference
 * (oracle/Makefile passes -I$(REF)/pgo/SC-A-LOAM/include) and compiled unmodified, together with
 * the reference's own nanoflann.hpp / KDTreeVectorOfVectorsAdaptor.h / tic_toc.h.  The libraries it
 * wants and this image lacks are replaced by oracle/standin/: a from-scratch Eigen stand-in that
 * reproduces Eigen 3.3's reduction order (see standin/Eigen/Dense), a two-member PCL stub and empty
 * OpenCV / cv_bridge / pcl_conversions headers (Scancontext.cpp uses nothing of those).
 * Outputs go to oracle/_ref/libref_sc_<order>.so (git-ignored; travels to the GPU box).
 *
 * Every function below only marshals plain buffers into the reference's types and calls the
 * reference's function; the arithmetic is the reference's.
 */
#include "scancontext/Scancontext.cpp"

#include <cstdint>
#include <cstring>
#include <sstream>

namespace {

MatrixXd to_mat(const double *p, int rows, int cols) {
  MatrixXd m(rows, cols);
  std::memcpy(m.data(), p, sizeof(double) * (size_t)rows * cols);
  return m;
}

pcl::PointCloud<SCPointType> to_cloud(const float *pts, size_t n, size_t stride_floats) {
  pcl::PointCloud<SCPointType> c;
  c.points.resize(n);
  for (size_t i = 0; i < n; i++) {
    c.points[i].x = pts[i * stride_floats + 0];
    c.points[i].y = pts[i * stride_floats + 1];
    c.points[i].z = pts[i * stride_floats + 2];
  }
  return c;
}

// the reference logs to std::cout (Scancontext.cpp:406,412); keep the test output clean
std::string g_last_log;
struct QuietCout {
  std::ostringstream sink;
  std::streambuf *old;
  QuietCout() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~QuietCout() {
    std::cout.rdbuf(old);
    g_last_log = sink.str();
  }
};

}  // namespace

extern "C" {

const char *ref_sc_build_info(void) {
  static char s[128];
  snprintf(s, sizeof(s), "packet=%d fma=%d predux34=%d", EIGEN_STANDIN_PACKET, EIGEN_STANDIN_FMA, EIGEN_STANDIN_PREDUX34);
  return s;
}

float ref_sc_xy2theta(float x, float y) { return xy2theta(x, y); }  // Scancontext.cpp:23-36

void ref_sc_circshift(const double *mat, int rows, int cols, int k, double *out) {  // Scancontext.cpp:39-59
  MatrixXd m = to_mat(mat, rows, cols);
  MatrixXd s = circshift(m, k);
  std::memcpy(out, s.data(), sizeof(double) * (size_t)rows * cols);
}

void ref_sc_make_scancontext(const float *pts, size_t n, size_t stride_floats, double *desc1200) {  // :151-195
  SCManager sc;
  auto cloud = to_cloud(pts, n, stride_floats);
  MatrixXd d = sc.makeScancontext(cloud);
  std::memcpy(desc1200, d.data(), sizeof(double) * 1200);
}

void ref_sc_ringkey(const double *desc, double *key20) {  // :198-211
  SCManager sc;
  MatrixXd d = to_mat(desc, 20, 60);
  MatrixXd k = sc.makeRingkeyFromScancontext(d);
  std::memcpy(key20, k.data(), sizeof(double) * 20);
}

void ref_sc_ringkey_f32(const double *desc, float *key20) {  // + eig2stdvec, :62-66
  SCManager sc;
  MatrixXd d = to_mat(desc, 20, 60);
  std::vector<float> v = eig2stdvec(sc.makeRingkeyFromScancontext(d));
  std::memcpy(key20, v.data(), sizeof(float) * 20);
}

void ref_sc_sectorkey(const double *desc, double *key60) {  // :214-227
  SCManager sc;
  MatrixXd d = to_mat(desc, 20, 60);
  MatrixXd k = sc.makeSectorkeyFromScancontext(d);
  std::memcpy(key60, k.data(), sizeof(double) * 60);
}

double ref_sc_dist_direct(const double *sc1, const double *sc2) {  // :69-90
  SCManager sc;
  MatrixXd a = to_mat(sc1, 20, 60), b = to_mat(sc2, 20, 60);
  return sc.distDirectSC(a, b);
}

int ref_sc_fast_align(const double *vkey1, const double *vkey2) {  // :93-113
  SCManager sc;
  MatrixXd a = to_mat(vkey1, 1, 60), b = to_mat(vkey2, 1, 60);
  return sc.fastAlignUsingVkey(a, b);
}

void ref_sc_distance(const double *sc1, const double *sc2, double *dist, int *shift) {  // :116-148
  SCManager sc;
  MatrixXd a = to_mat(sc1, 20, 60), b = to_mat(sc2, 20, 60);
  std::pair<double, int> r = sc.distanceBtnScanContext(a, b);
  *dist = r.first;
  *shift = r.second;
}

/* one query against n descriptors (n x 1200 doubles): the reference's pair function, nothing else */
void ref_sc_distances(const double *query, const double *descs, int64_t n, double *dist, int32_t *shift) {
  SCManager sc;
  MatrixXd q = to_mat(query, 20, 60);
  for (int64_t i = 0; i < n; i++) {
    MatrixXd e = to_mat(descs + i * 1200, 20, 60);
    std::pair<double, int> r = sc.distanceBtnScanContext(q, e);
    dist[i] = r.first;
    shift[i] = r.second;
  }
}

/* nq queries x n entries through the reference's own pair function, OpenMP over (query, block of 64 entries): the CPU
 * baseline bench.py reports beside the GPU number (`cpu_baseline.kind = "reference"`): Scancontext.cpp:116-148 itself, heap
 * allocations of its circshift / col() temporaries and all, on every host core.  dist, shift: [nq][n]. */
void ref_sc_distances_batch(const double *queries, int64_t nq, const double *descs, int64_t n, double *dist, int32_t *shift,
                            int nthreads) {
  const int64_t nb = (n + 63) / 64;
#pragma omp parallel for collapse(2) schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
  for (int64_t qi = 0; qi < nq; qi++)
    for (int64_t b = 0; b < nb; b++) {
      SCManager sc;
      MatrixXd q = to_mat(queries + qi * 1200, 20, 60);
      const int64_t i1 = (b + 1) * 64 < n ? (b + 1) * 64 : n;
      for (int64_t i = b * 64; i < i1; i++) {
        MatrixXd e = to_mat(descs + i * 1200, 20, 60);
        std::pair<double, int> r = sc.distanceBtnScanContext(q, e);
        dist[qi * n + i] = r.first;
        shift[qi * n + i] = r.second;
      }
    }
}

/* ---- the reference SCManager as a whole (Scancontext.h:57-122) ---- */
SCManager *ref_sc_create(void) { return new SCManager(); }
void ref_sc_destroy(SCManager *m) { delete m; }
void ref_sc_set_dist_thres(SCManager *m, double t) { m->setSCdistThres(t); }
int64_t ref_sc_size(const SCManager *m) { return (int64_t)m->polarcontexts_.size(); }

void ref_sc_add_points(SCManager *m, const float *pts, size_t n, size_t stride_floats) {  // :249-260
  auto cloud = to_cloud(pts, n, stride_floats);
  m->makeAndSaveScancontextAndKeys(cloud);
}

void ref_sc_add_descriptor(SCManager *m, const double *desc) {  // :236-246
  m->saveScancontextAndKeys(to_mat(desc, 20, 60));
}

void ref_sc_get(const SCManager *m, int64_t i, double *desc1200, float *ringkey20, double *sectorkey60) {
  if (desc1200) std::memcpy(desc1200, m->polarcontexts_[(size_t)i].data(), sizeof(double) * 1200);
  if (ringkey20) std::memcpy(ringkey20, m->polarcontext_invkeys_mat_[(size_t)i].data(), sizeof(float) * 20);
  if (sectorkey60) std::memcpy(sectorkey60, m->polarcontext_vkeys_[(size_t)i].data(), sizeof(double) * 60);
}

/* what the last detect call wrote to std::cout ("[Loop found] Nearest distance: ..." / "[Not loop] ...") */
const char *ref_sc_last_log(void) { return g_last_log.c_str(); }

/* detectLoopClosureID (:331-422); the log line's min_dist / nn_idx are not returned by the reference,
 * so only its return pair is exposed */
int ref_sc_detect_loop_closure(SCManager *m, float *yaw_diff_rad) {
  QuietCout q;
  std::pair<int, float> r = m->detectLoopClosureID();
  *yaw_diff_rad = r.second;
  return r.first;
}

int ref_sc_detect_between_session(SCManager *m, const float *key20, const double *desc, float *yaw_diff_rad) {  // :267-328
  QuietCout q;
  std::vector<float> key(key20, key20 + 20);
  MatrixXd d = to_mat(desc, 20, 60);
  std::pair<int, float> r = m->detectLoopClosureIDBetweenSession(key, d);
  *yaw_diff_rad = r.second;
  return r.first;
}
}
